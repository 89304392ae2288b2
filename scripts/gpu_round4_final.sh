#!/bin/bash
# round-4 end-of-round artifacts (one gpurun call): the default bench line, the rocprofv3 kernel trace of the same command (stats + per
# launch shape), the two HBM-traffic PMC passes, the attention PMC groups, the attention backward forms side by side, the other configs.
# Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py > gpurun_out/r4_bench_default_output.json 2> gpurun_out/r4_bench_default_output.err
cut -c1-300 gpurun_out/r4_bench_default_output.json
bash scripts/gpu_trace.sh r4_final > gpurun_out/r4_final_trace.txt 2>&1
bash scripts/gpu_trace_shapes.sh r4_final > /dev/null 2>&1
(python scripts/check_attn_sp.py --time-only 2>&1 | grep -v amdgpu.ids) > gpurun_out/r4_attn_bwd_forms.txt
(python scripts/bench_kernels.py quant 2>&1 | grep -v amdgpu.ids) > gpurun_out/r4_gemm_quant_cut.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r4_pmc_traffic.json | head -30
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
bash scripts/gpu_pmc_attention.sh r4 > /dev/null 2>&1
cat gpurun_out/r4_pmc_attention.txt
for spec in "cruller_small 2" "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  python bench.py --model $1 --batch $2 --graph-step off --no-cpu-baseline --no-roofline --no-host-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'step_mfma_frac', d['step_mfma_frac'], 'loss', d['loss'], '|', d['launch'])"
done > gpurun_out/r4_other_configs.txt 2>&1
cat gpurun_out/r4_other_configs.txt
