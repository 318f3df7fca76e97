set -u
mkdir -p gpurun_out/exp1
timeout 120 tools/bin/valu_mix > gpurun_out/exp1/valu_mix.txt 2>&1
cp cudacam_amd/libhipcanny.so /tmp/lib_wpb4.so
cp cudacam_amd/exp/libhipcanny_wpb1.so /tmp/lib_wpb1.so
cp cudacam_amd/exp/libhipcanny_wpb2.so /tmp/lib_wpb2.so
for w in 4 1 2; do
  cp /tmp/lib_wpb$w.so cudacam_amd/libhipcanny.so
  for s in 2 3 4; do
    HC_PIPE_SLOTS=$s timeout 200 python bench.py --rotate 1 --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp1/b_w${w}_s${s}.json 2> gpurun_out/exp1/b_w${w}_s${s}.err
    python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/exp1/b_w${w}_s${s}.json").read().strip().splitlines()[-1])
    r=j["roofline"]
    print("wpb $w slots $s: value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"bufs",j.get("output_buffers"),"hyst",j.get("hysteresis"))
except Exception as e:
    print("wpb $w slots $s: failed",e)
PY
  done
done
cp /tmp/lib_wpb4.so cudacam_amd/libhipcanny.so
