#!/bin/bash
# wave-quantisation remainder of the K = 1024 launches cut into slabs too (G_REM_SPLIT_MIN_NK 32 -> 16, slabs of >= 2 / 4 K tiles, modelled launch cost 14 -> 6 us):
# kernels and step, library against library
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
V=pixparse_amd/csrc/variants
NEW=pixparse_amd/csrc/libcruller_hip.so
( for rep in 1 2; do for lib in $NEW $V/libcruller_rem_16_2.so $V/libcruller_rem_16_4.so; do
    PIXPARSE_AMD_LIB=$lib PIXPARSE_AMD_SKIP_BUILD_CHECK=1 timeout 300 python scripts/bench_gemm4w.py k1024 2>&1 | grep -v amdgpu.ids | grep "qkv\|proj"
  done; done ) > gpurun_out/r6_rem_split_kernels.txt 2>&1
cat gpurun_out/r6_rem_split_kernels.txt
bash scripts/ab_libs_step.sh $NEW $V/libcruller_rem_16_2.so $V/libcruller_rem_16_4.so > gpurun_out/r6_rem_split_step_ab.txt 2>&1
cat gpurun_out/r6_rem_split_step_ab.txt
