"""How much of a gemm256 launch is NOT the K loop?  Times C[49152 x N] = A[49152 x K] . B^T for a sweep of K at fixed M, N
(768 or 3072 tiles of 256x256 = an exact number of rounds on 256 CUs) and fits  t = rounds * (a + b * K/64):
a = per-tile prologue + epilogue (exposed: one workgroup per CU, nothing overlaps it), b = time per 64-deep K tile.
Run on the GPU box:  python scripts/bench_epilogue.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops  # noqa: E402

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(layout, epi, M, N, Ks):
    rows = []
    for K in Ks:
        if layout == 'NT':
            x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
            out = torch.empty(M, N, dtype=BF16 if epi in (0, 1, 2) else torch.float32, device=dev)
            aux = torch.empty(M, N, dtype=BF16, device=dev) if epi == ops.EPI_BF16_GELU else None
            resid = torch.randn(M, N, device=dev) if epi == ops.EPI_F32_RESID else None
            bias = torch.randn(N, device=dev)
            fn = lambda: ops.linear_fwd(x, w, bias, out, epi, aux=aux, resid=resid)
        else:   # NN dgrad: out[M, N] = dy[M, K] @ w[K, N]
            dy = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(K, N, device=dev).to(BF16)
            out = torch.empty(M, N, dtype=BF16, device=dev)
            aux = torch.randn(M, N, device=dev).to(BF16) if epi == ops.EPI_BF16_DGELU else None
            fn = lambda: ops.linear_dgrad(dy, w, out, epi, aux=aux)
        ms = timeit(fn)
        rows.append((K, ms))
    rounds = (M // 256) * (N // 256) / 256.0
    # least squares t/rounds = a + b * (K/64)
    xs = [k / 64 for k, _ in rows]; ys = [ms * 1e3 / rounds for _, ms in rows]
    n = len(xs); sx, sy = sum(xs), sum(ys); sxx = sum(v * v for v in xs); sxy = sum(u * v for u, v in zip(xs, ys))
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    tf_loop = 2.0 * 256 * 256 * 64 * 256 / (b * 1e-6) / 1e12
    print(f'{layout} epi={epi} N={N}: per tile a = {a:6.2f} us (prologue+epilogue), b = {b:5.3f} us per K-tile ({tf_loop:6.0f} TF/s in the loop); '
          + ' '.join(f'K={k}:{ms:.3f}ms={2.0 * M * N * k / ms / 1e9:.0f}TF' for k, ms in rows), flush=True)


if __name__ == '__main__':
    pol = int(sys.argv[1]) if len(sys.argv) > 1 else 2     # 2 = 256x256 one workgroup per CU, 3 = 256x128 two per CU (same work per CU and round)
    print(f'policy {pol}', flush=True)
    hip.call('crl_gemm_set_policy', pol)
    M = 49152
    Ks = [128, 512, 1024, 2048, 4096]
    for epi in (ops.EPI_BF16, ops.EPI_BF16_GELU, ops.EPI_F32_RESID):
        run('NT', epi, M, 1024, Ks)
    run('NT', ops.EPI_BF16, M, 4096, [128, 512, 1024, 2048])
    run('NT', ops.EPI_BF16_GELU, M, 4096, [128, 512, 1024, 2048])
    for epi in (ops.EPI_BF16, ops.EPI_BF16_DGELU):
        run('NN', epi, M, 1024, Ks)
    run('NN', ops.EPI_BF16_DGELU, M, 4096, [128, 512, 1024, 2048])
    hip.call('crl_gemm_set_policy', 0)
