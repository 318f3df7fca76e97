set -u
mkdir -p gpurun_out/exp9
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp9/t_all.log 2>&1; tail -n 3 gpurun_out/exp9/t_all.log
for s in 8101 8102 8103 8104; do timeout 600 python tests/fuzz_parity.py 1500 $s > gpurun_out/exp9/f$s.log 2>&1 & done
wait
for s in 8101 8102 8103 8104; do tail -n 1 gpurun_out/exp9/f$s.log; done
run() { tag=$1; shift; timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp9/$tag.json 2> gpurun_out/exp9/$tag.err
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp9/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    bc={k:v["kernel_ms"] for k,v in (j.get("by_content") or {}).items()}
    print(t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),j.get("hysteresis"),bc)
except Exception as e: print(t,"failed",e)
PY
}
run nat --rotate 1
run rot
run noise --rotate 1 --kind noise
run bgr --rotate 1 --channels 3
run 4k --width 3840 --height 2160 --batch 256 --rotate 1
run 8k3 --width 7680 --height 4320 --channels 3 --per-channel --batch 16 --rotate 1 --unique 8
run modeo --rotate 1 --mode O
for nb in "1 0" "8 1"; do timeout 120 python tools/latency_trace.py $nb 2>&1 | tail -n 1; done
