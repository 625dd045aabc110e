#!/bin/bash
mkdir -p gpurun_out/r3c2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c2/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg2 --no-cpu --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r3c2/bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/r3c2/trace/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'gridhip' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 2 steps
names=[r['Kernel_Name'].split('(')[0][-48:] for r in rows]
# find index of last tile kernel and go back
idx=[i for i,r in enumerate(rows) if 'tile_grid_sorted' in r['Kernel_Name']]
a=idx[-3]+1; b=idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:7.1f} us  {r['Kernel_Name'].split('(')[0][-60:]}  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')}")
print(json.load(open('gpurun_out/r3c2/bench.json'))['ms_per_step'])
PY
