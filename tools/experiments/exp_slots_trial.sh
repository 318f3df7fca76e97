set -u
mkdir -p gpurun_out/exp14
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "third_slot" > gpurun_out/exp14/t.log 2>&1; tail -n 3 gpurun_out/exp14/t.log
run() { tag=$1; envs=$2; shift 2; env $envs timeout 300 python bench.py --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp14/$tag.json 2> gpurun_out/exp14/$tag.err
python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    j=json.loads(open(f"gpurun_out/exp14/{t}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"waves",j["buffers"].get("front_waves_per_workgroup"),"slots",j.get("pipeline_slots"),"bufs",j.get("output_buffers"))
except Exception as e: print(t,"failed",e)
PY
}
for b in 256 512; do
run b${b}_s2 HC_PIPE_SLOTS=2 --rotate 1 --batch $b --steps 200 --warmup 20 --out-buffers 3
run b${b}_s3 HC_PIPE_SLOTS=3 --rotate 1 --batch $b --steps 200 --warmup 20 --out-buffers 3
run b${b}_auto X=0 --rotate 1 --batch $b --steps 200 --warmup 20
done
run 8k1_auto X=0 --width 7680 --height 4320 --batch 32 --rotate 1 --unique 8 --steps 100 --warmup 20
run 8k1_s2 HC_PIPE_SLOTS=2 --width 7680 --height 4320 --batch 32 --rotate 1 --unique 8 --steps 100 --warmup 20 --out-buffers 3
