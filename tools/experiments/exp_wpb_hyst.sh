# hysteresis schedules beside a front kernel of one-wave workgroups (F8_WPB = 1) and of four (profiles/r03/experiments.md)
set -u
mkdir -p gpurun_out/exp3
cp cudacam_amd/libhipcanny.so /tmp/lib_wpb4.so
cp cudacam_amd/exp/libhipcanny_wpb1.so /tmp/lib_wpb1.so
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r=j["roofline"]
    print(sys.argv[1],": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"hyst",j.get("hysteresis"))
except Exception as e:
    print(sys.argv[1],": failed",e)
PY
}
run() {  # tag, env assignments, bench args
  local tag="$1" envs="$2"; shift 2
  env $envs timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed "$@" > gpurun_out/exp3/$tag.json 2> gpurun_out/exp3/$tag.err
  show "$tag [$envs]" gpurun_out/exp3/$tag.json
}
for w in 1 4; do
  cp /tmp/lib_wpb$w.so cudacam_amd/libhipcanny.so
  for sched in "X=0" "HC_HYST_MIXED_FROM=1" "HC_HYST_LISTS=1" "HC_HYST_LIST_FLOOR=512" "HC_HYST_MIXED_FROM=1 HC_HYST_LIST_FLOOR=512" "HC_HYST_LISTS=1 HC_HYST_LIST_FLOOR=512"; do
    t=$(echo "$sched" | tr ' =' '__')
    run nat_w${w}_$t "$sched" --rotate 1
    run bgr_w${w}_$t "$sched" --rotate 1 --channels 3
  done
  for sched in "X=0" "HC_HYST_MIXED_FROM=1" "HC_HYST_LISTS=1"; do
    t=$(echo "$sched" | tr ' =' '__')
    run rot_w${w}_$t "$sched"
  done
done
cp /tmp/lib_wpb4.so cudacam_amd/libhipcanny.so
