#!/bin/bash
# round 6: the stream variant for the first key block of a chain (no running-tile machinery): whole -m gpu suite, the backward alone and the step,
# library against library (firstoff = the same source with -DSPX_FIRST_VARIANT=0)
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
OFF=pixparse_amd/csrc/variants/libcruller_firstoff.so
NEW=pixparse_amd/csrc/libcruller_hip.so
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6_first_pytest.txt
cat gpurun_out/r6_first_pytest.txt
python - > gpurun_out/r6_first_kernels.txt 2>&1 <<'PY'
import os, subprocess, sys
code = r'''
import torch, sys, os
sys.path.insert(0, '.')
from pixparse_amd import hip, ops
hip.load(); dev = torch.device('cuda:0'); BF16 = torch.bfloat16
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, B, H, Nq, Nk in (('ViT', 8, 16, 6189, 6189), ('cross', 8, 16, 1023, 6189)):
    D = H * 64
    g = torch.Generator(device=dev).manual_seed(1)
    q = (torch.randn(B, Nq, D, generator=g, device=dev) * 0.125 * ops.LOG2E).to(BF16)
    k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(BF16) for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty_like(q); lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, 0.125, False, q_prescaled=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, Nq, device=dev)
    t = timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, 0.125, False, q_prescaled=True))
    print(f'{os.path.basename(os.environ.get("PIXPARSE_AMD_LIB", "product")):28s} {name:6s} backward (delta + stream + reduce): {t:8.1f} us', flush=True)
'''
for rep in range(3):
    for lib in ('pixparse_amd/csrc/variants/libcruller_firstoff.so', 'pixparse_amd/csrc/libcruller_hip.so'):
        env = dict(os.environ, PIXPARSE_AMD_LIB=lib, PIXPARSE_AMD_SKIP_BUILD_CHECK='1')
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        print(r.stdout.strip(), flush=True)
PY
cat gpurun_out/r6_first_kernels.txt
bash scripts/ab_libs_step.sh $OFF $NEW > gpurun_out/r6_first_step_ab.txt 2>&1
cat gpurun_out/r6_first_step_ab.txt
