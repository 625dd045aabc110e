set -u
export TMPDIR=/tmp
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  if [ $L = cur ]; then unset GRIDHIP_LIB; else export GRIDHIP_LIB=$PWD/tools/ab/libgridhip_$L.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/${L}_$C -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/${L}_$C.log 2>&1
  done
done
