#!/bin/bash
# Profile the bench workload on the GPU box: kernel-trace stats + separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md).
# usage: tools/profile.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py $ARGS > $D.log 2>&1
  echo "pmc $C rc=$?"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
