"""how much do the partial last rounds of the attention launches cost?  Same kernels at token counts that give whole / fractional numbers
of rounds of resident workgroups (B*H = 128 (b, h) pairs x ceil(N / 128) workgroups; 1024 / 768 / 512 slots for fwd / dQ / dK,dV)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixparse_amd import hip, ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B, H = 8, 16
D = H * 64
for rep in range(2):
    for N in (6144, 6189, 6272, 7168, 8192):
        qkv = torch.randn(B, N, 3 * D, device=dev).to(BF16)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        qp = (q.float() * (0.125 * ops.LOG2E)).to(BF16)
        o = torch.empty(B, N, D, dtype=BF16, device=dev); lse = torch.empty(B, H, N, device=dev)
        do = torch.randn(B, N, D, device=dev).to(BF16); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device=dev)
        t_f = timeit(lambda: ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True))
        bwd = lambda: ops.attn_bwd(qp, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, 0.125, False, q_prescaled=True)
        t = {}
        for name, mask in (('dkdv', 2), ('dq', 4)):
            hip.call('crl_attn_bwd_set_parts', mask)
            t[name] = timeit(bwd)
        hip.call('crl_attn_bwd_set_parts', 7)
        wg = B * H * ((N + 127) // 128)
        norm = (N / 6144.0) ** 2
        print(f'N={N:5d} workgroups {wg:5d}: fwd {t_f:6.3f} ms ({wg / 1024:5.2f} rounds, {t_f / norm:6.3f} per 6144^2)  dq {t["dq"]:6.3f} ({wg / 768:5.2f} rounds, {t["dq"] / norm:6.3f})  '
              f'dkdv {t["dkdv"]:6.3f} ({wg / 512:5.2f} rounds, {t["dkdv"] / norm:6.3f})', flush=True)
