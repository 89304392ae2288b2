"""single-pass attention backward: key blocks per workgroup (crl_attn_bwd_set_chain) -- time of the whole backward and dQ error against the
two-pass form, ViT shape of cfg-3 (B 8, H 16, N 6189) and the cross-attention shape (Nq 1023); run on the GPU box"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
from bench_kernels import timeit
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
chains = [int(c) for c in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 2, 3, 5, 7, 9, 13]
for (B, H, Nq, Nk) in ((8, 16, 6189, 6189), (8, 16, 1023, 6189)):
    D = H * 64
    g = torch.Generator(device=dev).manual_seed(1)
    q = (torch.randn(B, Nq, D, generator=g, device=dev) * (0.125 * ops.LOG2E)).to(BF16)
    k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(BF16) for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev); lse = torch.empty(B, H, Nq, device=dev); delta = torch.empty(2, B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, 0.125, False, q_prescaled=True)
    def run(mode, chain):
        dq, dk, dv = (torch.full((B, n, D), float('nan'), dtype=BF16, device=dev) for n in (Nq, Nk, Nk))
        hip.call('crl_attn_bwd_set_mode', mode); hip.call('crl_attn_bwd_set_chain', chain)
        f = lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, 0.125, False, q_prescaled=True)
        ms = timeit(f, iters=5)
        hip.call('crl_attn_bwd_set_mode', 0); hip.call('crl_attn_bwd_set_chain', 0)
        return ms, dq, dk, dv
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    ms2, dq2, dk2, dv2 = run(1, 0)
    print(f'B{B} H{H} Nq{Nq} Nk{Nk}: two-pass {ms2:.3f} ms')
    ref = run(2, 1)[2:]
    for rep in range(2):
        for c in chains:
            ms, dq, dk, dv = run(2, c)
            print(f'  chain {c:2d}: {ms:.3f} ms  dq vs two-pass {rel(dq, dq2):.2e}  dk/dv equal to chain 1: {torch.equal(dk, ref[0]) and torch.equal(dv, ref[1])}  finite {bool(torch.isfinite(dq.float()).all())}', flush=True)
