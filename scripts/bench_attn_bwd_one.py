"""ViT attention backward (delta + single pass + dQ reduce; B 8, H 16, N 6189, prescaled q) of the library named by PIXPARSE_AMD_LIB: us per call, three
rounds of 20.    python scripts/bench_attn_bwd_one.py [N]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
hip.load()
dev = torch.device('cuda:0')
B, H = 8, 16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6189
D = H * 64
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(B, N, 3 * D, generator=g, device=dev).to(torch.bfloat16)
qp = (qkv[:, :, :D].float() * 0.125 * ops.LOG2E).to(torch.bfloat16)
k, v = qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
o = torch.empty(B, N, D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, N, device=dev)
do = torch.randn(B, N, D, generator=g, device=dev).to(torch.bfloat16)
dqkv = torch.empty(B, N, 3 * D, dtype=torch.bfloat16, device=dev)
delta = torch.empty(2, B, H, N, device=dev)
ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
run = lambda: ops.attn_bwd(qp, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, 0.125, False, q_prescaled=True)
tag = os.path.basename(os.environ.get('PIXPARSE_AMD_LIB', 'product'))
res = []
for rnd in range(3):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f'{tag:28s} N {N}: ' + '  '.join(f'{t:7.1f} us' for t in res), flush=True)
