#!/bin/bash
# experiment: buffer epilogue (G_EPI_BUF) + templated ln_bwd: correctness, kernel A/B, step A/B
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/e1_tests.log 2>&1; tail -3 $O/e1_tests.log
bash scripts/ab_kernels.sh gemm256.hip gemm2x "pol=0" "-DG_EPI_BUF=0" "-DG_EPI_BUF=1" > $O/e1_ab_kernels.log 2>&1
bash scripts/ab_flags.sh gemm256.hip "-DG_EPI_BUF=0" "-DG_EPI_BUF=1" "-DG_EPI_BUF=0" "-DG_EPI_BUF=1" > $O/e1_ab_step.log 2>&1
cat $O/e1_ab_step.log
