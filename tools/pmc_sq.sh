set -u
OUT=gpurun_out/prof_sq
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu"
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py $ARGS > $D.log 2>&1
  echo "pmc $C rc=$?"
done
python3 - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/prof_sq/*/*/*counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tile_grid_sorted' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
