#!/bin/bash
# round 6, GELU epilogues that save gelu'(h) (fp16) for a multiply-only dgrad epilogue: the whole -m gpu suite, the K = 1024 encoder launches
# library against library, the step library against library (base = the pruned round-5 arithmetic)
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
BASE=pixparse_amd/csrc/variants/libcruller_r6base.so
NEW=pixparse_amd/csrc/libcruller_hip.so
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r6_gelu_pytest.txt
cat gpurun_out/r6_gelu_pytest.txt
for rep in 1 2; do for lib in $BASE $NEW; do PIXPARSE_AMD_LIB=$lib PIXPARSE_AMD_SKIP_BUILD_CHECK=1 timeout 300 python scripts/bench_gemm4w.py k1024 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r6_gelu_saved_derivative_kernels.txt 2>&1
cat gpurun_out/r6_gelu_saved_derivative_kernels.txt
bash scripts/ab_libs_step.sh $BASE $NEW > gpurun_out/r6_gelu_saved_derivative_step_ab.txt 2>&1
cat gpurun_out/r6_gelu_saved_derivative_step_ab.txt
