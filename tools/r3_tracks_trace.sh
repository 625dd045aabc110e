#!/bin/bash
mkdir -p gpurun_out/r3t
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3t/trace -- python3 $GRAFT_REPO_ROOT/tools/tracks_probe.py 64 > $GRAFT_REPO_ROOT/gpurun_out/r3t/out.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3t/trace/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'gridhip' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the probe runs 10 calls on the uniform stream, then 10 on tracks: take the 5th call of each by locating tile kernels
idx=[i for i,r in enumerate(rows) if 'tile_grid_sorted' in r['Kernel_Name']]
def call(k):
    a=idx[k-1]+1 if k>0 else 0; b=idx[k]+1
    return [(r['Kernel_Name'].split('(')[0][-40:], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows[a:b]]
for tag,k in (('uniform',5),('tracks64',15)):
    print(tag, ' '.join(f"{n.split('::')[-1][:22]}={t:.3f}" for n,t in call(k)))
PY
