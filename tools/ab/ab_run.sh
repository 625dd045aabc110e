# usage: tools/ab/ab_run.sh <outdir> <lib> [<lib> ...]   ("cur" = the library in the tree; others: tools/ab/libgridhip_<lib>.so)
# timings of cfg3 and cfg5 per library, twice in opposite orders (clock drift), through tools/sweep.py; extra option
# sets for the library in the tree come from $AB_SETS (space-separated sweep.py sets)
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
run() {  # <lib> <suffix>
  if [ $1 = cur ]; then unset GRIDHIP_LIB; else export GRIDHIP_LIB=$PWD/tools/ab/libgridhip_$1.so; fi
  python tools/sweep.py --reps 5 "" ${AB_SETS:-} > $O/$1_cfg3$2.log 2>&1
  python tools/sweep.py --workload cfg5 --reps 3 "" ${AB_SETS:-} > $O/$1_cfg5$2.log 2>&1
}
for L in "$@"; do run $L ""; done
for L in $(echo "$@" | tr ' ' '\n' | tac); do run $L _b; done
for L in "$@"; do for S in "" _b; do for W in cfg3 cfg5; do echo "== $L $W$S"; grep -v amdgpu.ids $O/${L}_$W$S.log | cut -c1-110; done; done; done
