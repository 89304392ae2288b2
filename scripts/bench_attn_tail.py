"""attention kernels at N = 6189 (6.125 / 12.25 workgroup rounds) against N = 6144 (6.0 / 12.0): how much does the last partial round cost?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, H = 8, 16
D = H * 64
for rep in range(2):
  for N in (6144, 6189, 6272, 6400):
    qkv = torch.randn(B, N, 3 * D, device=dev).to(BF16)
    o = torch.empty(B, N, D, dtype=BF16, device=dev); lse = torch.empty(B, H, N, device=dev)
    do = torch.randn(B, N, D, device=dev).to(BF16); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device=dev)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2*D], qkv[:, :, 2*D:]
    f = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, H, 0.125, False))
    bwd = lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2*D], dqkv[:, :, 2*D:], H, 0.125, False)
    t = {}
    for name, mask in (('dkdv', 2), ('dq', 4)):
        hip.call('crl_attn_bwd_set_parts', mask); t[name] = timeit(bwd)
    hip.call('crl_attn_bwd_set_parts', 7)
    w = (N / 6189.0) ** 2
    print(f'N={N}: fwd {f:.3f} ms ({f / w:.3f} per 6189^2 of work)  dkdv {t["dkdv"]:.3f} ({t["dkdv"] / w:.3f})  dq {t["dq"]:.3f} ({t["dq"] / w:.3f})')
