#!/bin/bash
mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/trace -- python3 $GRAFT_REPO_ROOT/tools/measure_all.py imaging > $GRAFT_REPO_ROOT/gpurun_out/r3i/out.jsonl 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-250 gpurun_out/r3i/out.jsonl | grep -v amdgpu
f=$(ls gpurun_out/r3i/trace/*/*_kernel_stats.csv | tail -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r['Name'][:78].ljust(78), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e6:8.3f} ms avg {float(r['TotalDurationNs'])/1e6:9.2f} ms total")
PY
