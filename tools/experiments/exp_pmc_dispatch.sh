set -u
R=$(pwd)
mkdir -p gpurun_out/exp6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/exp6/p1 -- python3 $R/bench.py --rotate 1 --steps 4 --warmup 3 --no-cpu-baseline --no-host-fed ${BENCH_ARGS:-} > /dev/null 2> $R/gpurun_out/exp6/p1.log
python3 $R/tools/pmc_per_dispatch.py $R/gpurun_out/exp6/p1 > $R/gpurun_out/exp6/per_dispatch_${TAG:-1080p}.txt 2>&1
rm -rf $R/gpurun_out/exp6/p1
cat $R/gpurun_out/exp6/per_dispatch_${TAG:-1080p}.txt
