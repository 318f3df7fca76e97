# rows per work item of k_front8 (runs that tile the 1080 rows evenly) with the run-time workgroup choice in effect
set -u
mkdir -p gpurun_out/exp17
for ch in 0 72 90 108 120 135 154 180 216 270 0; do
  timeout 300 python bench.py --rotate 1 --chunk $ch --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp17/c$ch.json 2> gpurun_out/exp17/c$ch.err
  python - $ch <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp17/c{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print("chunk",t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"waves",j["buffers"].get("front_waves_per_workgroup"))
except Exception as e: print(t,"failed",e)
PY
done
