#!/bin/bash
# round 6: ceiling of a cross-tile prefetch in the overlapped 4-wave GEMM: a timing-only build whose statements do not stage K tiles 0 and 1 at their entry
# (G4W_OPTS=nostage0: wrong results) against the product library, the K = 1024 launches + the step, alternating on one box
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
: > gpurun_out/r6_gemm_boundary.txt
for rep in 1 2; do
  for lib in pixparse_amd/csrc/libcruller_hip.so pixparse_amd/csrc/variants/libcruller_nostage0.so; do
    PIXPARSE_AMD_LIB=$lib PIXPARSE_AMD_SKIP_BUILD_CHECK=1 python scripts/bench_gemm4w.py k1024 2>&1 | grep -E "qkv|dgrad proj" | sed 's/| 8w[^|]*//' >> gpurun_out/r6_gemm_boundary.txt
  done
done
bash scripts/ab_libs_step.sh pixparse_amd/csrc/libcruller_hip.so pixparse_amd/csrc/variants/libcruller_nostage0.so >> gpurun_out/r6_gemm_boundary.txt 2>&1
cat gpurun_out/r6_gemm_boundary.txt
