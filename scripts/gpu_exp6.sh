#!/bin/bash
O=gpurun_out; mkdir -p $O
python scripts/bench_kernels.py gemm2x 2>&1 | grep -v amdgpu > $O/e6_pol03.log
bash scripts/ab_kernels.sh gemm256.hip gemm2x "pol=0" "-DG_CU_STAGGER=1" "-DG_CU_STAGGER=0" > $O/e6_ab_kernels.log 2>&1
bash scripts/ab_flags.sh gemm256.hip "-DG_CU_STAGGER=0" "-DG_CU_STAGGER=1" "-DG_CU_STAGGER=0" "-DG_CU_STAGGER=1" > $O/e6_ab_step.log 2>&1
cat $O/e6_ab_step.log
