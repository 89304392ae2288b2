#!/bin/bash
# round-6 first call: the whole -m gpu suite on the pruned library (+ the new production-geometry tests), the default bench line, and the
# same-box A/B of the serial order of the wave-quantisation remainder launch (G_REM_FIRST had no A/B on record)
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6_start_pytest.txt
cat gpurun_out/r6_start_pytest.txt
python bench.py > gpurun_out/r6_bench_start.json 2> gpurun_out/r6_bench_start.err
cut -c1-400 gpurun_out/r6_bench_start.json
bash scripts/ab_flags.sh gemm.hip "-DG_REM_FIRST=1" "-DG_REM_FIRST=0" "-DG_REM_FIRST=1" "-DG_REM_FIRST=0" "-DG_REM_FIRST=1" "-DG_REM_FIRST=0" > gpurun_out/r6_gemm_rem_order.txt 2>&1
cat gpurun_out/r6_gemm_rem_order.txt
