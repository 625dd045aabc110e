#!/bin/bash
mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/trace -- python3 $GRAFT_REPO_ROOT/tools/measure_all.py imaging > $GRAFT_REPO_ROOT/gpurun_out/r3i/out.jsonl 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-250 gpurun_out/r3i/out.jsonl | grep -v amdgpu
f=$(ls gpurun_out/r3i/trace/*/*_kernel_stats.csv | tail -1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3i/trace/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'mirror_kernel' in r['Kernel_Name']]
for which,tag in ((3,'simple_imaging resident, 10^6 vis'),(8,'w_cache_imaging resident, 10^6 vis')):
    a,b=idx[which],idx[which+1]
    t0=int(rows[a]['Start_Timestamp'])
    out=[]
    for r in rows[a:b]:
        n=r['Kernel_Name'].split('(')[0].replace('gridhip::','').replace('void ','')[:40]
        s_=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        if 'at::native' in n or 'elementwise' in n: continue
        if out and out[-1][0]==n: out[-1][2]+=d; out[-1][3]+=1
        else: out.append([n,s_,d,1])
    print('==',tag)
    for n,s_,d,c in out: print(f"{s_:8.1f} us {n:42s} x{c:<3d} {d:7.1f} us")
PY
