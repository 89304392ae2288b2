"""Single-pass attention backward (attn_bwd_fused_kernel + attn_dq_reduce_kernel) against the two-pass form and fp32 torch, then timing
of both at the cfg-3 encoder shape.   python scripts/check_attn_fused.py [--time-only]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixparse_amd import hip, ops

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(B, H, Nq, Nk, mode, seed=0, qkv=None):
    d, D = 64, H * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    if qkv is None:
        q = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
        k = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
        v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
        do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    else:
        q, k, v, do = qkv
    scale = d ** -0.5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False)
    dq, dk, dv = torch.full_like(q, float('nan')), torch.full_like(k, float('nan')), torch.full_like(v, float('nan'))
    delta = torch.empty(2, B, H, Nq, device=dev)
    hip.call('crl_attn_bwd_set_mode', mode)
    ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, scale, False)
    hip.call('crl_attn_bwd_set_mode', 0)
    torch.cuda.synchronize()
    return (q, k, v, do), o, lse, dq, dk, dv


def torch_ref(q, k, v, do, H):
    B, Nq, D = q.shape
    hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2).requires_grad_(True)
    Q, K, V = hd(q), hd(k), hd(v)
    O = torch.softmax(Q @ K.transpose(-1, -2) * 0.125, -1) @ V
    O.backward(do.float().reshape(B, Nq, H, 64).transpose(1, 2))
    back = lambda t: t.transpose(1, 2).reshape(B, -1, D)
    return back(Q.grad), back(K.grad), back(V.grad)


def main():
    if '--time-only' not in sys.argv:
        for (B, H, Nq, Nk) in [(1, 2, 300, 700), (2, 1, 64, 512), (1, 2, 100, 45), (1, 3, 1023, 1300), (1, 2, 6189, 6189), (1, 2, 1023, 6189)]:
            x, o, lse, dq2, dk2, dv2 = run(B, H, Nq, Nk, 1)
            _, _, _, dqf, dkf, dvf = run(B, H, Nq, Nk, 2, qkv=x)
            assert hip.query('crl_attn_bwd_ws_bytes', B, H, Nq, Nk, 0) == 0 or Nk >= 2048
            msg = f'B{B} H{H} Nq{Nq} Nk{Nk}: fused vs two-pass rel-L2 dq {rel(dqf, dq2):.2e} dk {rel(dkf, dk2):.2e} dv {rel(dvf, dv2):.2e}'
            if Nq * Nk <= 1023 * 1300:
                rq, rk, rv = torch_ref(*x, H)
                msg += f' | vs fp32: fused dq {rel(dqf, rq):.2e} dk {rel(dkf, rk):.2e} dv {rel(dvf, rv):.2e}; two-pass dq {rel(dq2, rq):.2e} dk {rel(dk2, rk):.2e} dv {rel(dv2, rv):.2e}'
            print(msg, flush=True)
            assert torch.isfinite(dqf.float()).all() and torch.isfinite(dkf.float()).all() and torch.isfinite(dvf.float()).all()
            assert rel(dqf, dq2) < 1e-2 and rel(dkf, dk2) < 5e-3 and rel(dvf, dv2) < 5e-3
            # determinism
            _, _, _, dqg, dkg, dvg = run(B, H, Nq, Nk, 2, qkv=x)
            assert torch.equal(dqg, dqf) and torch.equal(dkg, dkf) and torch.equal(dvg, dvf)
    # timing at the cfg-3 encoder shape
    B, H, N = 8, 16, 6189
    x, o, lse, *_ = run(B, H, N, N, 1)
    q, k, v, do = x
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, H, N, device=dev)
    for mode, name in ((1, 'two-pass'), (2, 'fused + reduce'), (1, 'two-pass'), (2, 'fused + reduce')):
        hip.call('crl_attn_bwd_set_mode', mode)
        f = lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, 0.125, False)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'{name:16s} {ms:.3f} ms per backward  ({8.0 * 64 * N * N * B * H / ms / 1e9:.0f} TFLOP/s algorithmic)', flush=True)
    hip.call('crl_attn_bwd_set_mode', 0)


if __name__ == '__main__':
    main()
