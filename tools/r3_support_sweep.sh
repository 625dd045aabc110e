# Support sweep at cfg3's shape: squares 17..32 in the tap-reusing kernel's own parts of the tap list (default) against round 2's
# sub-footprints (subfoot=1), and degrid2 at the same supports.  Output: gpurun_out/r3c/support_sweep.txt, degrid_sweep.txt
set -u
mkdir -p gpurun_out/r3c
rm -f gpurun_out/r3c/support_sweep.txt
for S in 15 16 17 19 21 23 25 27 29 31; do
  python tools/sweep.py --support $S --reps 2 "" "subfoot=1" 2>&1 | grep -v amdgpu.ids | sed "s/^/S=$S  /" | tee -a gpurun_out/r3c/support_sweep.txt
done
python tools/degrid_sweep.py 15 17 21 25 31 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3c/degrid_sweep.txt
python tools/degrid_sweep.py --subfoot 17 21 31 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3c/degrid_sweep.txt
