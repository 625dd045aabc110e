set -u
mkdir -p gpurun_out/r3e
python -m pytest tests -m gpu -q -x > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3e/pytest.log
python tools/sweep.py --support 17 --reps 3 "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/s17.txt
python tools/sweep.py --workload cfg5 --reps 3 "" "bigtile=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/cfg5_auto.txt
python tools/measure_all.py imaging > gpurun_out/r3e/imaging.jsonl 2>&1; cut -c1-330 gpurun_out/r3e/imaging.jsonl
python bench.py --workload cfg4 > gpurun_out/r3e/bench_cfg4.json 2> gpurun_out/r3e/bench_cfg4.err; echo "cfg4 rc=$?"; cut -c1-600 gpurun_out/r3e/bench_cfg4.json
python tools/measure_all.py aw > gpurun_out/r3e/aw.jsonl 2>&1; cat gpurun_out/r3e/aw.jsonl
