#!/bin/bash
# where the overlapped dGELU form loses its time: product library against timing-only variants without the aux loads / without the multiply
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
V=pixparse_amd/csrc/variants
( for rep in 1 2; do
    timeout 300 python scripts/bench_gemm4w.py dgelu4 2>&1 | grep -v amdgpu.ids
    PIXPARSE_AMD_GEMM_OVERLAP=0 timeout 300 python scripts/bench_gemm4w.py dgelu4 2>&1 | grep -v amdgpu.ids | sed 's/^/overlap off: /'
    for lib in $V/libcruller_drop_auxload.so $V/libcruller_drop_auxmath.so $V/libcruller_drop_auxload_auxmath.so; do
      PIXPARSE_AMD_LIB=$lib PIXPARSE_AMD_SKIP_BUILD_CHECK=1 timeout 300 python scripts/bench_gemm4w.py dgelu4 2>&1 | grep -v amdgpu.ids | grep dgelu
    done
  done ) > gpurun_out/r6_ovlg_drops.txt 2>&1
cat gpurun_out/r6_ovlg_drops.txt
