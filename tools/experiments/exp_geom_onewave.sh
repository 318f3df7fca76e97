# hysteresis workgroup shapes beside one-wave front workgroups (HC_HYST_GEOM = rows x waves; product: 32x2)
set -u
mkdir -p gpurun_out/exp16
run() { tag=$1; envs=$2; shift 2; env $envs timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp16/$tag.json 2> gpurun_out/exp16/$tag.err
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp16/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"waves",j["buffers"].get("front_waves_per_workgroup"),j["hysteresis"])
except Exception as e: print(t,"failed",e)
PY
}
for g in 32x2 32x1 32x4 32x2; do
  run nat_$g HC_HYST_GEOM=$g --rotate 1
  run rot_$g HC_HYST_GEOM=$g
done
