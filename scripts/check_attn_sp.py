"""Single-pass attention backward with a prescaled q (the training configuration): the hand-placed stream (mode 2) against the C++ form of
the same algorithm (mode 3), the two-pass kernels (mode 1) and fp32 torch; then timing at the cfg-3 encoder and cross-attention shapes.
    python scripts/check_attn_sp.py [--time-only] [--modes 1,2]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixparse_amd import hip, ops

dev = torch.device('cuda:0')
BF16 = torch.bfloat16
SCALE = 0.125
C = SCALE * ops.LOG2E


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make(B, H, Nq, Nk, seed=0, strided=False):
    D = H * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    if strided and Nq == Nk:      # q | k | v as column blocks of one [B, N, 3D] projection output, like the ViT blocks
        qkv = torch.randn(B, Nq, 3 * D, generator=g, device=dev)
        qkv[:, :, :D] *= C
        qkv = qkv.to(BF16)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    else:
        q = (torch.randn(B, Nq, D, generator=g, device=dev) * C).to(BF16)
        k = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
        v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, SCALE, False, q_prescaled=True)
    return q, k, v, do, o, lse


def bwd(x, H, mode):
    q, k, v, do, o, lse = x
    B, Nq, D = do.shape
    dq, dk, dv = (torch.full((B, n, D), float('nan'), dtype=BF16, device=dev) for n in (Nq, k.shape[1], k.shape[1]))
    delta = torch.empty(2, B, H, Nq, device=dev)
    hip.call('crl_attn_bwd_set_mode', mode)
    try:
        ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, SCALE, False, q_prescaled=True)
    finally:
        hip.call('crl_attn_bwd_set_mode', 0)
    torch.cuda.synchronize()
    return dq, dk, dv


def torch_ref(x, H):
    q, k, v, do, o, lse = x
    B, Nq, D = do.shape
    hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
    qproj = (hd(q) / C).requires_grad_(True)
    K, V = hd(k).requires_grad_(True), hd(v).requires_grad_(True)
    z = (qproj * C) @ K.transpose(-1, -2) * math.log(2.0)
    (torch.softmax(z, -1) @ V).backward(hd(do))
    back = lambda t: t.transpose(1, 2).reshape(B, -1, D)
    return back(qproj.grad), back(K.grad), back(V.grad)


def stamps(nwg, Nq):
    """diagnostic build (-DSPX_STAMPS): shader cycles and 100 MHz ticks each workgroup spent in the asm statement"""
    import ctypes
    lib = hip.load()
    if not hasattr(lib, 'crl_debug_spx_stamps'):
        return
    n = min(nwg, 4096)
    buf = (ctypes.c_ulonglong * (2 * n))()
    lib.crl_debug_spx_stamps(buf, 2 * n)
    cyc = sorted(buf[2 * i] for i in range(n))
    tick = sorted(buf[2 * i + 1] for i in range(n))
    T = (Nq + 63) // 64 + 1
    med_c, med_t = cyc[n // 2], tick[n // 2]
    print(f'  stamps over {n} workgroups: median {med_c} cycles = {med_c / T:.0f} per tile pass = {med_c / T / 80:.1f} per MFMA; '
          f'{med_t / 100:.1f} us -> in-kernel clock {med_c / med_t * 0.1:.3f} GHz (min {cyc[0]}, max {cyc[-1]} cycles)', flush=True)
    if hasattr(lib, 'crl_debug_spx_trace') and T < 128:
        import statistics
        tr = (ctypes.c_ulonglong * (6 * 65 * 128))()
        lib.crl_debug_spx_trace(tr)
        seg = {k: [] for k in ('top wait', 'barrier', 'slot 0', 'slot 1', 'slot 2', 'slot 3')}
        for wg in range(64):
            st = lambda k, p: tr[(k * 65 + wg) * 128 + p]
            for p in range(3, T - 1):           # pass p's stamps sit in row p + 1 (written at the top of the next pass)
                s0, s1, s2, s3, s4, s5 = (st(k, p + 1) for k in range(6))
                prev_end = st(5, p)
                if min(s0, s1, s2, s3, s4, s5, prev_end) == 0:
                    continue
                seg['top wait'].append(s0 - prev_end); seg['barrier'].append(s1 - s0); seg['slot 0'].append(s2 - s1)
                seg['slot 1'].append(s3 - s2); seg['slot 2'].append(s4 - s3); seg['slot 3'].append(s5 - s4)
        pro = {k: [] for k in ('wait for DMA / K,V loads', 'K^T + first fragments', 'first S / dP')}
        for wg in range(64):
            a, b, c, d = (tr[(k * 65 + wg) * 128 + 0] for k in (2, 3, 4, 5))
            if min(a, b, c, d) > 0:
                pro['wait for DMA / K,V loads'].append(b - a); pro['K^T + first fragments'].append(c - b); pro['first S / dP'].append(d - c)
        if pro['first S / dP']:
            print('  prologue (cycles, median / max over 64 workgroups of the first round): ' + ', '.join(f'{k} {statistics.median(v):.0f} / {max(v):.0f}' for k, v in pro.items()), flush=True)
        print('  per-pass trace (cycles, median / p90 over 64 workgroups x passes): ' +
              ', '.join(f'{k} {statistics.median(v):.0f} / {sorted(v)[int(len(v) * 0.9)]:.0f}' for k, v in seg.items() if v), flush=True)


def main():
    modes = [1, 2]
    if '--modes' in sys.argv:
        modes = [int(m) for m in sys.argv[sys.argv.index('--modes') + 1].split(',')]
    bad = 0
    if '--time-only' not in sys.argv:
        for (B, H, Nq, Nk, strided) in [(1, 1, 64, 256, False), (1, 2, 300, 700, False), (2, 1, 64, 512, False), (1, 2, 100, 45, False), (1, 3, 1023, 1300, False),
                                        (2, 2, 577, 577, True), (1, 2, 6189, 6189, True), (1, 2, 1023, 6189, False)]:
            x = make(B, H, Nq, Nk, seed=Nq + Nk, strided=strided)
            dq1, dk1, dv1 = bwd(x, H, 1)
            dq2, dk2, dv2 = bwd(x, H, 2)
            msg = f'B{B} H{H} Nq{Nq} Nk{Nk}: asm vs two-pass dq {rel(dq2, dq1):.2e} dk {rel(dk2, dk1):.2e} dv {rel(dv2, dv1):.2e}'
            if '--ref' in sys.argv:
                dq3, dk3, dv3 = bwd(x, H, 3)
                msg += f' | asm vs C++ form: dq {rel(dq2, dq3):.1e} dk {rel(dk2, dk3):.1e} dv {rel(dv2, dv3):.1e} (bit-equal: {torch.equal(dq2, dq3)}, {torch.equal(dk2, dk3)}, {torch.equal(dv2, dv3)})'
            if Nq * Nk <= 1023 * 1300:
                rq, rk, rv = torch_ref(x, H)
                msg += f' | vs fp32: asm dq {rel(dq2, rq):.2e} dk {rel(dk2, rk):.2e} dv {rel(dv2, rv):.2e}; two-pass dq {rel(dq1, rq):.2e} dk {rel(dk1, rk):.2e} dv {rel(dv1, rv):.2e}'
            fin = all(bool(torch.isfinite(t.float()).all()) for t in (dq2, dk2, dv2))
            ok = fin and rel(dq2, dq1) < 1e-2 and rel(dk2, dk1) < 5e-3 and rel(dv2, dv1) < 5e-3
            dq4, dk4, dv4 = bwd(x, H, 2)
            det = torch.equal(dq4, dq2) and torch.equal(dk4, dk2) and torch.equal(dv4, dv2)
            print(('OK   ' if ok and det else 'FAIL ') + msg + ('' if det else ' NOT DETERMINISTIC') + ('' if fin else ' NON-FINITE'), flush=True)
            bad += 0 if (ok and det) else 1
    for (B, H, Nq, Nk, name) in [(8, 16, 6189, 6189, 'ViT-L self'), (8, 16, 1023, 6189, 'cross')]:
        x = make(B, H, Nq, Nk, strided=True)
        q, k, v, do, o, lse = x
        dq, dk, dv = torch.empty_like(do), torch.empty_like(k), torch.empty_like(v)
        if Nq == Nk:
            dqkv = torch.empty(B, Nq, 3 * H * 64, dtype=BF16, device=dev)
            dq, dk, dv = dqkv[:, :, :H * 64], dqkv[:, :, H * 64:2 * H * 64], dqkv[:, :, 2 * H * 64:]
        delta = torch.empty(2, B, H, Nq, device=dev)
        for mode in modes + modes:
            hip.call('crl_attn_bwd_set_mode', mode)
            f = lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, SCALE, False, q_prescaled=True)
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
            print(f'{name:12s} mode {mode}: {ms:.3f} ms per backward  ({8.0 * 64 * Nq * Nk * B * H / ms / 1e9:.0f} TFLOP/s algorithmic)', flush=True)
        hip.call('crl_attn_bwd_set_mode', 0)
        stamps(B * H * ((Nk + 255) // 256), Nq)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
