#!/bin/bash
# round-3 end-of-round artifacts (one gpurun call): kernel A/Bs as text, the rocprofv3 kernel trace (stats + per launch shape), the two PMC
# passes, the default bench line.  Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python scripts/bench_kernels.py quant 2>&1 | grep -v amdgpu.ids) > gpurun_out/r3_gemm_quant_cut.txt
(python scripts/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids) > gpurun_out/r3_attn_prescaled_q.txt
(python scripts/bench_kernels.py dec 2>&1 | grep -v amdgpu.ids) > gpurun_out/r3_gemm_decoder_shapes.txt
bash scripts/gpu_trace.sh r3_final > gpurun_out/r3_final_trace.txt 2>&1
bash scripts/gpu_trace_shapes.sh r3_final > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r3_pmc_traffic.json | head -30
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
python bench.py > gpurun_out/r3_bench_default_output.json 2> gpurun_out/r3_bench_default_output.err
cut -c1-400 gpurun_out/r3_bench_default_output.json
# timing-only experiment last (it rebuilds the library on this box): half of the LDS fragment reads of the dK/dV pass
(bash scripts/ab_kernel_flags.sh "" "-DATT_TIMING_HALF_LDS_BWD" "" "-DATT_TIMING_HALF_LDS_BWD" 2>&1) > gpurun_out/r3_attn_bwd_half_lds_experiment.txt
tail -8 gpurun_out/r3_attn_bwd_half_lds_experiment.txt
