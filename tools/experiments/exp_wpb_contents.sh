# WPB = 1 against the product's 4 on the other workloads; HC_FRONT_FILL for small pipelined batches (profiles/r03/experiments.md)
set -u
mkdir -p gpurun_out/exp2
cp cudacam_amd/libhipcanny.so /tmp/lib_wpb4.so
cp cudacam_amd/exp/libhipcanny_wpb1.so /tmp/lib_wpb1.so
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r=j["roofline"]
    bc={k:(v["frames_per_s"],v["kernel_ms"]) for k,v in (j.get("by_content") or {}).items()}
    print(sys.argv[1],": value",j["value"],"ms/step",j["ms_per_step"],"kernel_ms",r["kernel_ms"],"hyst_ms",r.get("hyst_expand_ms"),"continued",j.get("hysteresis",{}).get("continued"),bc)
except Exception as e:
    print(sys.argv[1],": failed",e)
PY
}
for w in 4 1; do
  cp /tmp/lib_wpb$w.so cudacam_amd/libhipcanny.so
  timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/rot_w$w.json 2> gpurun_out/exp2/rot_w$w.err; show "wpb $w rotation" gpurun_out/exp2/rot_w$w.json
  timeout 300 python bench.py --rotate 1 --kind noise --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/noise_w$w.json 2> gpurun_out/exp2/noise_w$w.err; show "wpb $w noise" gpurun_out/exp2/noise_w$w.json
  timeout 300 python bench.py --rotate 1 --width 3840 --height 2160 --batch 256 --steps 30 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/4k_w$w.json 2> gpurun_out/exp2/4k_w$w.err; show "wpb $w 4K" gpurun_out/exp2/4k_w$w.json
  timeout 300 python bench.py --rotate 1 --width 640 --height 480 --batch 4096 --steps 30 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/vga_w$w.json 2> gpurun_out/exp2/vga_w$w.err; show "wpb $w 640x480" gpurun_out/exp2/vga_w$w.json
  timeout 300 python bench.py --rotate 1 --channels 3 --steps 30 --warmup 8 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/bgr_w$w.json 2> gpurun_out/exp2/bgr_w$w.err; show "wpb $w BGR" gpurun_out/exp2/bgr_w$w.json
  for b in 64 256; do
    timeout 300 python bench.py --rotate 1 --batch $b --steps 60 --warmup 10 --no-cpu-baseline --no-host-fed > gpurun_out/exp2/b${b}_w$w.json 2> gpurun_out/exp2/b${b}_w$w.err; show "wpb $w batch $b" gpurun_out/exp2/b${b}_w$w.json
  done
  for fill in 3072 2048 1536 1024; do
    for nb in "1 0" "4 1" "8 1" "14 1"; do
      echo -n "wpb $w fill $fill: "; HC_FRONT_FILL=$fill timeout 120 python tools/latency_trace.py $nb 2>&1 | tail -1
    done
  done
done
cp /tmp/lib_wpb4.so cudacam_amd/libhipcanny.so
