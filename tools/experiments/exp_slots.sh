set -u
mkdir -p gpurun_out/exp4
for s in 2 3 4; do
  HC_PIPE_SLOTS=$s timeout 300 python bench.py --width 7680 --height 4320 --channels 3 --per-channel --batch 16 --rotate 1 --unique 8 --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp4/8k3_s$s.json 2> gpurun_out/exp4/8k3_s$s.err
  HC_PIPE_SLOTS=$s timeout 300 python bench.py --width 7680 --height 4320 --batch 32 --rotate 1 --unique 8 --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp4/8k1_s$s.json 2> gpurun_out/exp4/8k1_s$s.err
  HC_PIPE_SLOTS=$s timeout 300 python bench.py --channels 3 --rotate 1 --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp4/bgr_s$s.json 2> gpurun_out/exp4/bgr_s$s.err
  HC_PIPE_SLOTS=$s timeout 300 python bench.py --mode O --rotate 1 --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp4/modeo_s$s.json 2> gpurun_out/exp4/modeo_s$s.err
  for t in 8k3 8k1 bgr modeo; do python - $t $s <<'PY'
import json,sys
t,s=sys.argv[1:3]
try:
    j=json.loads(open(f"gpurun_out/exp4/{t}_s{s}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(t,"slots",s,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"bufs",j.get("output_buffers"),j.get("hysteresis"))
except Exception as e: print(t,s,"failed",e)
PY
  done
done
