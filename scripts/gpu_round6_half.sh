#!/bin/bash
# round 6: half-chip contraction split (decoder fc2 / fc1 dgrad / LM-head dgrad on the 4-wave 256x256 kernel in two contraction slices): gemm tests, kernel
# timings against the previous commit's library (128x128 kernel), step A/B
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
OLD=pixparse_amd/csrc/variants/libcruller_nohalf.so
NEW=pixparse_amd/csrc/libcruller_hip.so
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "gemm or decoder or cruller or train" 2>&1 | tail -4 > gpurun_out/r6_half_pytest.txt
cat gpurun_out/r6_half_pytest.txt
python - > gpurun_out/r6_half_kernels.txt 2>&1 <<'PY'
import os, subprocess, sys
code = r'''
import torch, sys, os
sys.path.insert(0, '.')
from pixparse_amd import hip, ops
hip.load(); dev = torch.device('cuda:0'); BF16 = torch.bfloat16
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tag = os.path.basename(os.environ.get("PIXPARSE_AMD_LIB", "product"))
M = 8184
for name, N, K, kind in (('dec fc2 + resid', 1024, 4096, 'resid'), ('dec fc1 dgrad', 1024, 4096, 'nn'), ('LM-head dgrad', 1024, 50304, 'nn'), ('dec out_proj + resid', 1024, 1024, 'resid'), ('dec qkv dgrad', 1024, 3072, 'nn')):
    if kind == 'resid':
        x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16); b = torch.randn(N, device=dev); y = torch.randn(M, N, device=dev)
        t = timed(lambda: ops.linear_fwd(x, w, b, y, ops.EPI_F32_RESID, resid=y))
    else:
        dy = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(K, N, device=dev).to(BF16); out = torch.empty(M, N, dtype=BF16, device=dev)
        t = timed(lambda: ops.linear_dgrad(dy, w, out))
    print(f'{tag:24s} {name:22s} {M}x{N}x{K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s', flush=True)
'''
for rep in range(2):
    for lib in ('pixparse_amd/csrc/variants/libcruller_nohalf.so', 'pixparse_amd/csrc/libcruller_hip.so'):
        env = dict(os.environ, PIXPARSE_AMD_LIB=lib, PIXPARSE_AMD_SKIP_BUILD_CHECK='1')
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        print(r.stdout.strip(), flush=True)
PY
cat gpurun_out/r6_half_kernels.txt
bash scripts/ab_libs_step.sh $OLD $NEW > gpurun_out/r6_half_step_ab.txt 2>&1
cat gpurun_out/r6_half_step_ab.txt
