# k_front8 compiled with other instruction-scheduling options of the AMDGPU back end (front8.hip only; product flags: none)
set -u
mkdir -p gpurun_out/exp10
cp cudacam_amd/libhipcanny.so /tmp/lib_base.so
for t in base memclause nopost trackers maxilp base; do
  if [ $t = base ]; then cp /tmp/lib_base.so cudacam_amd/libhipcanny.so; else cp cudacam_amd/exp/lib_$t.so cudacam_amd/libhipcanny.so; fi
  for cfg in "nat --rotate 1" "natnp --rotate 1 --no-pipeline"; do
    tag=$(echo $cfg | cut -d' ' -f1); args=$(echo $cfg | cut -s -d' ' -f2-)
    timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed $args > gpurun_out/exp10/${tag}_$t.json 2> gpurun_out/exp10/${tag}_$t.err
    python - $tag $t <<'PY'
import json,sys
t,v=sys.argv[1:3]
try:
    j=json.loads(open(f"gpurun_out/exp10/{t}_{v}.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print(v,t,": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"])
except Exception as ex: print(v,t,"failed",ex)
PY
  done
done
cp /tmp/lib_base.so cudacam_amd/libhipcanny.so
