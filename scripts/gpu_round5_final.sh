#!/bin/bash
# round-5 end-of-round artifacts (one gpurun call): the default bench line, the rocprofv3 kernel trace of the same command (stats + per launch
# shape), the two HBM-traffic PMC passes, the GEMM and forward-attention PMC groups (4-wave against 8-wave kernel at 8192^3), the yardstick against the vendor libraries, the other configs.
# Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/r5_bench_default_output.json 2> gpurun_out/r5_bench_default_output.err
cut -c1-300 gpurun_out/r5_bench_default_output.json
bash scripts/gpu_trace.sh r5_final > gpurun_out/r5_final_trace.txt 2>&1
bash scripts/gpu_trace_shapes.sh r5_final > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r5_pmc_traffic.json | head -30
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
bash scripts/gpu_pmc_gemm.sh r5 > /dev/null 2>&1
cat gpurun_out/r5_pmc_gemm.txt
bash scripts/gpu_pmc_attn_fwd.sh r5 > /dev/null 2>&1
cat gpurun_out/r5_pmc_attn_fwd.txt
bash scripts/gpu_yardstick.sh > /dev/null 2>&1
for spec in "cruller_small 2" "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  python bench.py --model $1 --batch $2 --graph-step off --no-cpu-baseline --no-roofline --no-host-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'step_mfma_frac', d['step_mfma_frac'], 'loss', d['loss'], '|', d['launch'])"
done > gpurun_out/r5_other_configs.txt 2>&1
cat gpurun_out/r5_other_configs.txt
