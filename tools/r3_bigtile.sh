set -u
mkdir -p gpurun_out/r3d
python tools/reserve_cus_probe.py cfg3 > gpurun_out/r3d/reserve_cfg3.txt 2>&1; tail -14 gpurun_out/r3d/reserve_cfg3.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "knobs or sorted or convgrid2_matches" > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3d/pytest.log
for WL in cfg3 cfg5; do
  python tools/sweep.py --workload $WL --reps 3 "" "bigtile=1" "bigtile=1,wtable=1" "bigtile=1,wtable=2" 2>&1 | grep -v amdgpu.ids | sed "s/^/$WL  /" | tee -a gpurun_out/r3d/bigtile.txt
done
for S in 9 13 17 21 25 31; do
  python tools/sweep.py --support $S --reps 2 "" "bigtile=1" 2>&1 | grep -v amdgpu.ids | sed "s/^/S=$S  /" | tee -a gpurun_out/r3d/bigtile.txt
done
