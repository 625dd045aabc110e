set -u
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "support or subfoot or sorted or convgrid2_matches or knobs or golden or degrid" > gpurun_out/r3c/pytest_parity.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r3c/pytest_parity.log
python tools/reserve_cus_probe.py cfg3 > gpurun_out/r3c/reserve_cfg3.txt 2>&1; tail -12 gpurun_out/r3c/reserve_cfg3.txt
for S in 15 16 17 19 21 23 25 27 29 31; do
  python tools/sweep.py --support $S --reps 2 "" "subfoot=1" 2>&1 | grep -v amdgpu.ids | sed "s/^/S=$S  /" | tee -a gpurun_out/r3c/support_sweep.txt
done
