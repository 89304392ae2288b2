#!/bin/bash
# round 6: the overlapped dGELU form of the 4-wave GEMM (bit-identity + kernel timings + step A/B against the round's base library) and the
# pricing micro-benchmark of the two-waves-per-SIMD attention-backward forms (scripts/gen_ubench_bwd_forms.py)
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
BASE=pixparse_amd/csrc/variants/libcruller_r6base.so
NEW=pixparse_amd/csrc/libcruller_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -5 > gpurun_out/r6_ovlg_pytest.txt
cat gpurun_out/r6_ovlg_pytest.txt
( for rep in 1 2; do timeout 300 python scripts/bench_gemm4w.py k1024 2>&1 | grep -v amdgpu.ids | grep "dgelu\|gelu"; PIXPARSE_AMD_GEMM_OVERLAP=0 timeout 300 python scripts/bench_gemm4w.py k1024 2>&1 | grep "dgelu" | sed 's/^/overlap off: /'; done
  timeout 300 python scripts/bench_gemm4w.py decdgelu 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6_ovlg_kernels.txt 2>&1
cat gpurun_out/r6_ovlg_kernels.txt
timeout 300 ./scripts/ubench_bwd_forms.bin 1500 > gpurun_out/r6_ubench_bwd_forms.txt 2>&1
timeout 300 ./scripts/ubench_bwd_forms.bin 1500 >> gpurun_out/r6_ubench_bwd_forms.txt 2>&1
cat gpurun_out/r6_ubench_bwd_forms.txt
bash scripts/ab_libs_step.sh $BASE $NEW > gpurun_out/r6_ovlg_step_ab.txt 2>&1
cat gpurun_out/r6_ovlg_step_ab.txt
