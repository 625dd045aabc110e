#!/bin/bash
# which number of w-groups is fastest, by shape and stream size (tools/sweep.py); "default" = the library's own choice
S='"" "wgroups=8" "wgroups=4" "wgroups=2" "wgroups=1"'
run() { tag=$1; shift; eval timeout -k 10 200 python tools/sweep.py "$@" $S 2>&1 | grep -v amdgpu.ids | sed "s/^/$tag /"; }
for NV in 1500000 3000000 6000000 12500000 25000000 50000000; do run "cfg3_n=$NV" --nvis $NV --reps 5; done
run "cfg3" --reps 3
run "cfg5" --workload cfg5 --reps 3
run "cfg2" --workload cfg2 --reps 7
run "cfg2_1e7" --workload cfg2 --nvis 10000000 --reps 5
run "cfg3_7x7" --support 7 --reps 3
run "cfg3_31x31_2e7" --support 31 --nvis 20000000 --reps 3
