#!/bin/bash
# after the last change to attention.hip: the two HBM-traffic PMC passes (bench.py quotes them only while the kernel source's sha1 matches), then the
# default bench line (which now carries roofline.traffic) and the kernel trace of the same command
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r5_pmc_traffic.json | head -8
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp gpurun_out/r5_pmc_traffic.json profiles/r5_pmc_traffic.json
python bench.py > gpurun_out/r5_bench_default_output.json 2> gpurun_out/r5_bench_default_output.err
cut -c1-300 gpurun_out/r5_bench_default_output.json
bash scripts/gpu_trace.sh r5_final > gpurun_out/r5_final_trace.txt 2>&1
bash scripts/gpu_trace_shapes.sh r5_final > /dev/null 2>&1
