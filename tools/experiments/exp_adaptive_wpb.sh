set -u
mkdir -p gpurun_out/exp13
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/exp13/t_all.log 2>&1; tail -n 3 gpurun_out/exp13/t_all.log
run() { tag=$1; shift; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp13/$tag.json 2> gpurun_out/exp13/$tag.err
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp13/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"waves",j["buffers"].get("front_waves_per_workgroup"),"slots",j.get("pipeline_slots"),"cont",j["hysteresis"]["continued"])
except Exception as e: print(t,"failed",e)
PY
}
run nat --rotate 1
run rot
run noise --rotate 1 --kind noise
run bgr --rotate 1 --channels 3
run 4k --width 3840 --height 2160 --batch 256 --rotate 1
run vga --rotate 1 --width 640 --height 480 --batch 4096
run 8k1 --width 7680 --height 4320 --batch 32 --rotate 1 --unique 8
run b256 --rotate 1 --batch 256
run nat2 --rotate 1
run rot2
