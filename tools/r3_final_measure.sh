#!/bin/bash
# Round-3 measurement pass on one MI355X: rocprofv3 kernel statistics + PMC traffic for cfg3 and the cfg5 share, kernel
# statistics for cfg4, phase profiles (tuning build), bench lines of every workload, the single-GPU pieces of the
# multi-GPU prediction, the secondary configurations.  Outputs under gpurun_out/; tools/collect_profiles.py
# copies the summaries into profiles/.
set -u
OUT=gpurun_out/r3m
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile.sh r03_cfg3 > $OUT/profile_cfg3.log 2>&1; echo "profile cfg3 rc=$?"
bash tools/profile.sh r03_cfg5 --workload cfg5 > $OUT/profile_cfg5.log 2>&1; echo "profile cfg5 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_cfg4/trace -- python3 bench.py --workload cfg4 --no-cpu --steps 5 --warmup 2 > $OUT/prof_cfg4.log 2>&1; echo "trace cfg4 rc=$?"
python tools/phase_profile.py --workload=cfg3 > $OUT/phase_cfg3.log 2>&1; echo "phase cfg3 rc=$?"
python tools/phase_profile.py --workload=cfg5 "" bigtile=2 > $OUT/phase_cfg5.log 2>&1; echo "phase cfg5 rc=$?"
for WL in cfg3 cfg2 cfg5 cfg4; do
  python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; echo "bench $WL rc=$?"
done
python bench.py --dist core --no-cpu > $OUT/bench_cfg3_core.json 2> $OUT/bench_cfg3_core.err; echo "bench core rc=$?"
# single-GPU pieces of the multi-GPU prediction: the step with 32 CUs reserved, at each GPU count's share of the stream
for NV in 100000000 50000000 25000000 12500000; do
  python tools/sweep.py --nvis $NV --reps 3 "" "yield_cus=64" "reserve_cus=32" 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg3 nvis=$NV  /" | tee -a $OUT/multigpu_pieces.txt
done
python tools/sweep.py --workload cfg5 --reps 3 "" "yield_cus=64" "reserve_cus=32" 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg5  /" | tee -a $OUT/multigpu_pieces.txt
python tools/measure_all.py cfg2 core degrid plan host > $OUT/secondary.jsonl 2>&1; echo "secondary rc=$?"
