# red-black schedule of the per-tile hysteresis launches (HC_HYST_RB=1, the default under test) against the Jacobi schedule (=0)
set -u
mkdir -p gpurun_out/exp11
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp11/t_all.log 2>&1; tail -n 3 gpurun_out/exp11/t_all.log
for s in 9101 9102 9103 9104; do timeout 600 python tests/fuzz_parity.py 1500 $s > gpurun_out/exp11/f$s.log 2>&1 & done
wait
for s in 9101 9102 9103 9104; do tail -n 1 gpurun_out/exp11/f$s.log; done
run() { rb=$1; tag=$2; shift 2; HC_HYST_RB=$rb timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp11/${tag}_$rb.json 2> gpurun_out/exp11/${tag}_$rb.err
python - $tag $rb <<'PY'
import json,sys
t,rb=sys.argv[1:3]
try:
    j=json.loads(open(f"gpurun_out/exp11/{t}_{rb}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    bc={k:v["kernel_ms"] for k,v in (j.get("by_content") or {}).items()}
    print("rb",rb,t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),j.get("hysteresis"),bc)
except Exception as e: print(t,rb,"failed",e)
PY
}
for rb in 0 1 0 1; do
  run $rb nat --rotate 1
  run $rb rot
done
for rb in 0 1; do
  run $rb noise --rotate 1 --kind noise
  run $rb bgr --rotate 1 --channels 3
  run $rb modeo --rotate 1 --mode O
  run $rb vga --rotate 1 --width 640 --height 480 --batch 4096
  run $rb natnp --rotate 1 --no-pipeline
  run $rb b256 --rotate 1 --batch 256
  for nb in "1 0" "8 1"; do echo -n "rb $rb: "; HC_HYST_RB=$rb timeout 120 python tools/latency_trace.py $nb 2>&1 | tail -n 1; done
done
