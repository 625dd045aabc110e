#!/bin/bash
# aw gridders after a change: tests, two cfg4 bench lines, kernel trace
mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_aw.py -m gpu -x -q > gpurun_out/r3h/aw.log 2>&1; echo "aw tests rc=$?"; tail -2 gpurun_out/r3h/aw.log
for i in 1 2; do
  python bench.py --workload cfg4 --no-cpu > gpurun_out/r3h/cfg4_$i.json 2>/dev/null
  python -c "
import json; r=json.load(open('gpurun_out/r3h/cfg4_$i.json')); print('cfg4', round(r['value'],1), 'Mvis/s', round(r['ms_per_step'],3), 'ms; build phase', round(r['roofline']['build_ms']['median'],3))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3h/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg4 --no-cpu --steps 5 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r3h/trace/*/*_kernel_stats.csv | tail -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
