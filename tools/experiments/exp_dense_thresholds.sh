# dense-path thresholds (half-lanes per window of 768 that send the next window to the wave-wide NMS / back), border rows no longer counted
set -u
mkdir -p gpurun_out/exp7
for th in "512 384" "400 300" "320 240" "256 192"; do
  set -- $th
  for cfg in "nat --rotate 1" "rot" "noise --rotate 1 --kind noise"; do
    set -- $th; e=$1; l=$2
    tag=$(echo $cfg | cut -d' ' -f1); args=$(echo $cfg | cut -s -d' ' -f2-)
    HC_DENSE_ENTER=$e HC_DENSE_LEAVE=$l timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed $args > gpurun_out/exp7/${tag}_$e.json 2> gpurun_out/exp7/${tag}_$e.err
    python - $tag $e $l <<'PY'
import json,sys
t,e,l=sys.argv[1:4]
try:
    j=json.loads(open(f"gpurun_out/exp7/{t}_{e}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    bc={k:v["kernel_ms"] for k,v in (j.get("by_content") or {}).items()}
    print(t,"enter",e,"leave",l,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],bc)
except Exception as ex: print(t,e,"failed",ex)
PY
  done
done
