# PMC traffic of the tile kernel (FETCH_SIZE x 2 + WRITE_SIZE per launch), quick form: tools/profile.sh is the full pass
set -u
OUT=gpurun_out/r3t
mkdir -p $OUT
export TMPDIR=/tmp
for WL in ${*:-cfg3 cfg5}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$OUT/${WL}_$C
    rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --workload $WL --no-cpu --steps 3 --warmup 1 > $D.log 2>&1
  done
  python3 - <<PY
import csv,glob
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    v=[]
    for f in glob.glob("$OUT/${WL}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_grid_sorted" in r["Kernel_Name"] and r["Counter_Name"]==c: v.append(float(r["Counter_Value"]))
    out[c]=sum(v)/max(len(v),1)
print("$WL  fetch(x2) %.2f GB  write %.2f GB  total %.2f GB" % (out["FETCH_SIZE"]*2048/1e9, out["WRITE_SIZE"]*1024/1e9, (out["FETCH_SIZE"]*2048+out["WRITE_SIZE"]*1024)/1e9))
PY
done 2>&1 | tee $OUT/summary.txt
