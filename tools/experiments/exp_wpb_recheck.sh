set -u
mkdir -p gpurun_out/exp12
cp cudacam_amd/libhipcanny.so /tmp/lib_wpb4.so
cp cudacam_amd/exp/libhipcanny_wpb1.so cudacam_amd/libhipcanny.so
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_abi_cpu.py > gpurun_out/exp12/t_wpb1.log 2>&1; tail -n 3 gpurun_out/exp12/t_wpb1.log
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=j["roofline"]
    print(sys.argv[1],": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"))
except Exception as e: print(sys.argv[1],"failed",e)
PY
}
for rep in 1 2; do for w in 4 1; do
  if [ $w = 4 ]; then cp /tmp/lib_wpb4.so cudacam_amd/libhipcanny.so; else cp cudacam_amd/exp/libhipcanny_wpb1.so cudacam_amd/libhipcanny.so; fi
  timeout 300 python bench.py --rotate 1 --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp12/nat_w${w}_$rep.json 2>/dev/null; show "wpb $w nat" gpurun_out/exp12/nat_w${w}_$rep.json
  timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp12/rot_w${w}_$rep.json 2>/dev/null; show "wpb $w rot" gpurun_out/exp12/rot_w${w}_$rep.json
done; done
cp /tmp/lib_wpb4.so cudacam_amd/libhipcanny.so
