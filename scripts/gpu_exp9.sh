#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/e9_tests.log 2>&1; tail -3 $O/e9_tests.log
bash scripts/ab_flags.sh gemm.hip "-DG_REM_SPLIT=0" "-DG_REM_SPLIT=1" "-DG_REM_SPLIT=0" "-DG_REM_SPLIT=1" > $O/e9_ab_step.log 2>&1
cat $O/e9_ab_step.log
