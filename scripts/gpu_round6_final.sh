#!/bin/bash
# round-6 end-of-round artifacts (one gpurun call): the two HBM-traffic PMC passes first (bench.py quotes them only while attention.hip's sha1 matches),
# the default bench line, the rocprofv3 kernel trace of the same command (stats + per launch shape), the other configs, cross-process determinism.
# Everything lands in gpurun_out/; the summaries are copied to profiles/ by hand.
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-host-leg --no-peak > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r6_pmc_traffic.json | head -12
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp gpurun_out/r6_pmc_traffic.json profiles/r6_pmc_traffic.json
python bench.py > gpurun_out/r6_bench_default_output.json 2> gpurun_out/r6_bench_default_output.err
cut -c1-400 gpurun_out/r6_bench_default_output.json
bash scripts/gpu_trace.sh r6_final > gpurun_out/r6_final_trace.txt 2>&1
tail -50 gpurun_out/r6_final_trace.txt
bash scripts/gpu_trace_shapes.sh r6_final > /dev/null 2>&1
for spec in "cruller_small 2" "cruller_base_960x640 8" "cruller_large_6layers 2"; do
  set -- $spec
  python bench.py --model $1 --batch $2 --graph-step off --no-cpu-baseline --no-roofline --no-host-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 batch $2:', d['value'], 'docs/s', d['ms_per_step'], 'ms/step', 'step_mfma_frac', d['step_mfma_frac'], 'loss', d['loss'], '|', d['launch'])"
done > gpurun_out/r6_other_configs.txt 2>&1
cat gpurun_out/r6_other_configs.txt
bash scripts/gpu_determinism.sh "PIXPARSE_AMD_GEMM_BIG=2" "PIXPARSE_AMD_FUSE_DBIAS=0" > gpurun_out/r6_determinism.txt 2>&1
cat gpurun_out/r6_determinism.txt
