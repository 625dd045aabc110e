set -u
mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_aw.py -m gpu -x -q > gpurun_out/r3g/pytest_aw.log 2>&1; echo "aw tests rc=$?"; tail -3 gpurun_out/r3g/pytest_aw.log
export GRIDHIP_LIB=$PWD/ska-sdp-accelerate-gridding_amd/lib/libgridhip_tuning.so
for D in 0 4 1 2 3 0 4; do
  timeout -k 10 300 python bench.py --workload cfg4 --no-cpu --steps 6 --warmup 2 --opt dbg=$D > gpurun_out/r3g/cfg4_d$D.json 2> gpurun_out/r3g/cfg4_d$D.err
  python - <<PY
import json
try:
    r=json.load(open("gpurun_out/r3g/cfg4_d$D.json"))
    print("dbg=$D  build_ms median", round(r["roofline"]["build_ms"]["median"],3), " value", round(r["value"],1), " frac", r["roofline"].get("frac"))
except Exception as e:
    print("dbg=$D failed", e, open("gpurun_out/r3g/cfg4_d$D.err").read()[-300:])
PY
done 2>&1 | tee gpurun_out/r3g/summary.txt
