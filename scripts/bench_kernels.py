"""micro-benchmarks of the hot kernels on random data (HIP events); run on the GPU box"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def gemm_case(name, layout, M, N, K, epi=ops.EPI_BF16, policy=0):
    hip.call('crl_gemm_set_policy', policy)
    if layout == 'NT':
        x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16 if epi in (0, 1, 2) else torch.float32, device=dev)
        aux = torch.empty(M, N, dtype=BF16, device=dev) if epi == ops.EPI_BF16_GELU else None
        resid = out if epi == ops.EPI_F32_RESID else None
        bias = torch.randn(N, device=dev)
        fn = lambda: ops.linear_fwd(x, w, bias, out, epi, aux=aux, resid=resid)
    elif layout == 'NN':
        dy = torch.randn(M, N, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, K, dtype=BF16, device=dev)
        aux = torch.randn(M, K, device=dev).to(BF16) if epi == ops.EPI_BF16_DGELU else None
        fn = lambda: ops.linear_dgrad(dy, w, out, epi, aux=aux)
    else:
        dy = torch.randn(M, N, device=dev).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
        dw = torch.zeros(N, K, device=dev)
        fn = lambda: ops.linear_wgrad(dy, x, dw, True)
    ms = timeit(fn)
    print(f'{name:34s} {layout} M={M} N={N} K={K} epi={epi} pol={policy}: {ms:7.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s', flush=True)
    hip.call('crl_gemm_set_policy', 0)

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'gemm'
    if which == 'gemm2x':   # automatic plan (256x256 one-per-CU + remainder) vs the 256x128 two-per-CU kernel, step shapes, interleaved
        M = 49512
        for rep in range(2):
            for pol in (0, 2):      # (policy 3 = the 256x128 two-per-CU kernel left the library in round 6: the forced 256x256 kernels instead)
                gemm_case('qkv', 'NT', M, 3072, 1024, policy=pol)
                gemm_case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, policy=pol)
                gemm_case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU, policy=pol)
                gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, policy=pol)
                gemm_case('dec kv', 'NT', M, 2048, 1024, policy=pol)
                gemm_case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, policy=pol)
                gemm_case('dgrad fc1', 'NN', M, 4096, 1024, policy=pol)
                gemm_case('dgrad qkv', 'NN', M, 3072, 1024, policy=pol)
                gemm_case('dgrad proj', 'NN', M, 1024, 1024, policy=pol)
                gemm_case('lm head', 'NT', 8184, 50304, 1024, policy=pol)
                gemm_case('square 8192', 'NT', 8192, 8192, 8192, policy=pol)
        sys.exit(0)
    if which == 'quant':    # the wave-quantisation cut at several cost constants, encoder step shapes, interleaved
        M = 49512
        for rep in range(2):
            for c in (1.0, 0.0, -1.0):      # the model, always cut the last partial round, never cut
                hip.call('crl_gemm_set_quant_cost', c)
                print(f'--- quant cost {c}')
                gemm_case('qkv', 'NT', M, 3072, 1024)
                gemm_case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID)
                gemm_case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU)
                gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID)
                gemm_case('dec kv', 'NT', M, 2048, 1024)
                gemm_case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU)
                gemm_case('dgrad fc1', 'NN', M, 4096, 1024)
                gemm_case('dgrad qkv', 'NN', M, 3072, 1024)
                gemm_case('dgrad proj', 'NN', M, 1024, 1024)
                gemm_case('dgrad dec kv', 'NN', M, 2048, 1024)
        sys.exit(0)
    if which == 'enc2x':    # encoder shapes with the epilogue-heavy K = 1024 GEMMs: automatic plan (256x256 persistent + cut) vs the 256x128 two-per-CU kernel
        M = 49512
        for rep in range(2):
            for pol in (0, 2):      # (policy 3 = the 256x128 two-per-CU kernel left the library in round 6: the forced 256x256 kernels instead)
                gemm_case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, policy=pol)
                gemm_case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU, policy=pol)
                gemm_case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, policy=pol)
                gemm_case('dgrad proj', 'NN', M, 1024, 1024, policy=pol)
                gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, policy=pol)
                gemm_case('qkv', 'NT', M, 3072, 1024, policy=pol)
        sys.exit(0)
    if which == 'dec':      # decoder-side shapes (M = 8 x 1023 rows): automatic plan vs the 256x128 two-per-CU kernel vs forced 256x256
        M = 8184
        for rep in range(2):
            for pol in (0, 2):
                gemm_case('qkv', 'NT', M, 3072, 1024, policy=pol)
                gemm_case('out resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, policy=pol)
                gemm_case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU, policy=pol)
                gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, policy=pol)
                gemm_case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, policy=pol)
                gemm_case('dgrad fc1', 'NN', M, 4096, 1024, policy=pol)
                gemm_case('dgrad qkv', 'NN', M, 3072, 1024, policy=pol)
                gemm_case('dgrad proj', 'NN', M, 1024, 1024, policy=pol)
        sys.exit(0)
    if which == 'gemm':
        for pol in (2, 1):
            gemm_case('square 8192', 'NT', 8192, 8192, 8192, policy=pol)
            gemm_case('square 4096', 'NT', 4096, 4096, 4096, policy=pol)
        M = 49512
        for pol in (1, 2):
            gemm_case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, policy=pol)
            gemm_case('qkv', 'NT', M, 3072, 1024, policy=pol)
            gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, policy=pol)
            gemm_case('dgrad proj', 'NN', M, 1024, 1024, policy=pol)
            gemm_case('wgrad proj', 'TN', M, 1024, 1024, policy=pol)
            gemm_case('wgrad fc1', 'TN', M, 4096, 1024, policy=pol)
            gemm_case('dec qkv', 'NT', 8184, 3072, 1024, policy=pol)
            gemm_case('dec fc1', 'NT', 8184, 4096, 1024, ops.EPI_BF16_GELU, policy=pol)
            gemm_case('dec out resid', 'NT', 8184, 1024, 1024, ops.EPI_F32_RESID, policy=pol)
        gemm_case('qkv', 'NT', M, 3072, 1024)
        gemm_case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID)
        gemm_case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU)
        gemm_case('fc1 plain', 'NT', M, 4096, 1024)
        gemm_case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID)
        gemm_case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU)
        gemm_case('dgrad fc1', 'NN', M, 4096, 1024)
        gemm_case('dgrad qkv', 'NN', M, 3072, 1024)
        gemm_case('wgrad fc1', 'TN', M, 4096, 1024)
        gemm_case('wgrad proj', 'TN', M, 1024, 1024)
        gemm_case('lm head', 'NT', 8184, 50304, 1024)
    else:
        B, H, N = 8, 16, 6189
        D = H * 64
        qkv = torch.randn(B, N, 3 * D, device=dev).to(BF16)
        o = torch.empty(B, N, D, dtype=BF16, device=dev); lse = torch.empty(B, H, N, device=dev)
        do = torch.randn(B, N, D, device=dev).to(BF16); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device=dev)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2*D], qkv[:, :, 2*D:]
        ms = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, H, 0.125, False))
        fl = 4.0 * N * N * D * B
        print(f'attn fwd  {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s (algorithmic)')
        qp = (q.float() * (0.125 * ops.LOG2E)).to(BF16)
        for rep in range(2):
            ms = timeit(lambda: ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True))
            print(f'attn fwd, prescaled q  {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s (algorithmic)')
            ms = timeit(lambda: ops.attn_fwd(q, k, v, o, lse, H, 0.125, False))
            print(f'attn fwd               {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s (algorithmic)')
        ms = timeit(lambda: ops.attn_bwd(qp, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2*D], dqkv[:, :, 2*D:], H, 0.125, False, q_prescaled=True))
        print(f'attn bwd, prescaled q  {ms:7.3f} ms')
        ms = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2*D], dqkv[:, :, 2*D:], H, 0.125, False))
        print(f'attn bwd  {ms:7.3f} ms {2*fl/ms/1e9:7.1f} TF/s (algorithmic 2x fwd; executed 3.5x)')
        bwd = lambda: ops.attn_bwd(q, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2*D], dqkv[:, :, 2*D:], H, 0.125, False)
        for rep in range(2):   # the two backward passes separately (crl_attn_bwd_set_parts)
            for name, mask in (('dkdv', 2), ('dq', 4)):
                hip.call('crl_attn_bwd_set_parts', mask)
                ms = timeit(bwd, 10)
                print(f'attn bwd part {name:12s} {ms:7.3f} ms')
        hip.call('crl_attn_bwd_set_parts', 7)
