set -u
mkdir -p gpurun_out/exp5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "third_slot or ring or pipelin" > gpurun_out/exp5/t.log 2>&1; tail -5 gpurun_out/exp5/t.log
run() { tag=$1; shift; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp5/$tag.json 2> gpurun_out/exp5/$tag.err
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp5/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"bufs",j.get("output_buffers"),j.get("hysteresis"))
except Exception as e: print(t,"failed",e)
PY
}
run 8k3 --width 7680 --height 4320 --channels 3 --per-channel --batch 16 --rotate 1 --unique 8
run 8k1 --width 7680 --height 4320 --batch 32 --rotate 1 --unique 8
run 4k --width 3840 --height 2160 --batch 256 --rotate 1
run nat --rotate 1
run rot
run bgr --rotate 1 --channels 3
run modeo --rotate 1 --mode O
run vga --rotate 1 --width 640 --height 480 --batch 4096
