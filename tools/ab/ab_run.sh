# usage: tools/ab/ab_run.sh <outdir> <lib> [<lib> ...]   ("cur" = the library in the tree)
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  if [ $L = cur ]; then unset GRIDHIP_LIB; else export GRIDHIP_LIB=$PWD/tools/ab/libgridhip_$L.so; fi
  python tools/sweep.py --reps 5 "" > $O/${L}_cfg3.log 2>&1
  python tools/sweep.py --workload cfg5 --reps 3 "" > $O/${L}_cfg5.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/${L}_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/${L}_$C.log 2>&1
  done
done
# second timing round, reverse order (clock drift)
for L in $(echo "$@" | tr ' ' '\n' | tac); do
  if [ $L = cur ]; then unset GRIDHIP_LIB; else export GRIDHIP_LIB=$PWD/tools/ab/libgridhip_$L.so; fi
  python tools/sweep.py --reps 5 "" > $O/${L}_cfg3_b.log 2>&1
done
grep -h -o "kernel *[0-9.]* " $O/*cfg3.log | head -0
for L in "$@"; do echo "$L cfg3: $(grep -o 'prepass.*Mvis/s   ' $O/${L}_cfg3.log) | b: $(grep -o 'kernel *[0-9.]*' $O/${L}_cfg3_b.log | head -1) | cfg5: $(grep -o 'kernel *[0-9.]*' $O/${L}_cfg5.log | head -1)"; done
