"""Same-box alternation: the 4-wave one-wave-per-SIMD GEMM (gemm4w.hip) against the 8-wave kernel (gemm256.hip) and the vendor GEMM, random data,
back-to-back launches.   python scripts/bench_gemm4w.py [quick]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, layout, M, N, K, epi=ops.EPI_BF16, vendor=False, policy=0, rounds=2):
    if layout == 'NT':
        x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16 if epi in (0, 1, 2) else torch.float32, device=dev)
        aux = torch.empty(M, N, dtype=BF16, device=dev) if epi == ops.EPI_BF16_GELU else None
        resid = out if epi == ops.EPI_F32_RESID else None
        bias = torch.randn(N, device=dev)
        fn = lambda: ops.linear_fwd(x, w, bias, out, epi, aux=aux, resid=resid)
        wt = w.t(); o2 = torch.empty(M, N, dtype=BF16, device=dev)
        fv = lambda: torch.matmul(x, wt, out=o2)
    elif layout == 'NN':
        dy = torch.randn(M, N, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, K, dtype=BF16, device=dev)
        aux = torch.randn(M, K, device=dev).to(BF16) if epi == ops.EPI_BF16_DGELU else None
        fn = lambda: ops.linear_dgrad(dy, w, out, epi, aux=aux)
        fv = lambda: torch.matmul(dy, w, out=out)
    else:
        dy = torch.randn(M, N, device=dev).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
        dw = torch.zeros(N, K, device=dev)
        fn = lambda: ops.linear_wgrad(dy, x, dw, True)
        o2 = torch.empty(N, K, dtype=BF16, device=dev); dyt = dy.t()
        fv = lambda: torch.matmul(dyt, x, out=o2)
    flops = 2.0 * M * N * K
    n = max(10, int(200 * min(1.0, 1.1e12 / flops))) if not QUICK else 10
    hip.call('crl_gemm_set_policy', policy)
    for r in range(rounds):
        row = []
        for big in (0, 1):
            hip.call('crl_gemm_set_big_kernel', big)
            ms = timed(fn, n)
            row.append(f'{"4w" if big else "8w"} {ms * 1000:8.1f} us {flops / ms / 1e9:7.1f} TF/s')
        if vendor:
            ms = timed(fv, n)
            row.append(f'vendor {ms * 1000:8.1f} us {flops / ms / 1e9:7.1f} TF/s')
        print(f'{name:26s} {layout} {M}x{N}x{K} epi {epi} pol {policy} | ' + ' | '.join(row), flush=True)
    hip.call('crl_gemm_set_policy', 0); hip.call('crl_gemm_set_big_kernel', 2)


if __name__ == '__main__':
    M = 49512
    if len(sys.argv) > 1 and sys.argv[1] == 'k1024':    # the epilogue-heavy K = 1024 launches of an encoder block (+ the two residual ones)
        tag = os.path.basename(os.environ.get('PIXPARSE_AMD_LIB', 'default'))
        case(f'{tag} qkv', 'NT', M, 3072, 1024, rounds=1)
        case(f'{tag} proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, rounds=1)
        case(f'{tag} fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU, rounds=1)
        case(f'{tag} fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, rounds=1)
        case(f'{tag} dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, rounds=1)
        case(f'{tag} dgrad proj', 'NN', M, 1024, 1024, rounds=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dgelu4':    # the fc2 dgrad shape: plain store against the dGELU (multiply by the saved derivative) epilogue, with / without the wave-quantisation cut
        tag = os.path.basename(os.environ.get('PIXPARSE_AMD_LIB', 'product'))
        for pol in (0, 2):
            case(f'{tag} dgrad fc2 plain', 'NN', M, 1024, 4096, ops.EPI_BF16, policy=pol, rounds=1)
            case(f'{tag} dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, policy=pol, rounds=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'decdgelu':    # the decoder's fc2 dgrad + dGELU (8184 rows: two rounds of tiles, classic forms only)
        case('dec dgrad fc2 dgelu', 'NN', 8184, 1024, 4096, ops.EPI_BF16_DGELU, rounds=2)
        case('dec fc1 gelu', 'NT', 8184, 4096, 1024, ops.EPI_BF16_GELU, rounds=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'epi2x':    # the epilogue-heavy launches: automatic plan against the forced 256x256 kernels (policy 3, the 256x128 two-per-CU kernel, left the library in round 6)
        for pol in (0, 2, 0, 2):
            case(f'fc1 gelu pol {pol}', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU, policy=pol, rounds=1)
            case(f'dgrad fc2 dgelu pol {pol}', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU, policy=pol, rounds=1)
            case(f'proj resid pol {pol}', 'NT', M, 1024, 1024, ops.EPI_F32_RESID, policy=pol, rounds=1)
            case(f'fc2 resid pol {pol}', 'NT', M, 1024, 4096, ops.EPI_F32_RESID, policy=pol, rounds=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'decoder':    # the decoder's M = 8184 GEMMs with 1024 output columns (128 tiles of 256 x 256: the automatic plan takes the 128 x 128 kernel)
        Md = 8184
        for pol in (0, 2, 0, 2):
            case(f'dec proj pol {pol}', 'NT', Md, 1024, 1024, ops.EPI_F32_RESID, policy=pol, rounds=1)
            case(f'dec fc2 pol {pol}', 'NT', Md, 1024, 4096, ops.EPI_F32_RESID, policy=pol, rounds=1)
            case(f'dec dgrad K1024 pol {pol}', 'NN', Md, 1024, 1024, policy=pol, rounds=1)
            case(f'dec dgrad K4096 pol {pol}', 'NN', Md, 4096, 1024, policy=pol, rounds=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'wide':     # very wide outputs: automatic plan (256x128 two-per-CU kernel) against the forced 256x256 kernels
        for pol in (0, 2):
            case(f'lm head pol {pol}', 'NT', 8184, 50304, 1024, policy=pol, rounds=1)
            case(f'square 8192 pol {pol}', 'NT', 8192, 8192, 8192, policy=pol, rounds=1)
            case(f'lm head dgrad pol {pol}', 'NN', 8184, 1024, 50304, policy=pol, rounds=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'one':      # variant runs (scripts/ab_g4w.sh): the main-loop-bound case + one K = 1024 case
        tag = os.path.basename(os.environ.get('PIXPARSE_AMD_LIB', 'default'))
        case(f'{tag} square 8192', 'NT', 8192, 8192, 8192, policy=2)
        case(f'{tag} proj K=1024', 'NT', M, 1024, 1024, policy=2)
        case(f'{tag} dgrad fc1 NN', 'NN', M, 4096, 1024, rounds=1)
        case(f'{tag} wgrad fc1 TN', 'TN', M, 4096, 1024, rounds=1)
        sys.exit(0)
    case('square 8192', 'NT', 8192, 8192, 8192, vendor=True, policy=2)
    case('proj plain K=1024', 'NT', M, 1024, 1024, vendor=True, policy=2)
    case('fc2 plain K=4096', 'NT', M, 1024, 4096, vendor=True, policy=2)
    case('qkv', 'NT', M, 3072, 1024, vendor=True)
    case('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID)
    case('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU)
    case('fc2 resid', 'NT', M, 1024, 4096, ops.EPI_F32_RESID)
    case('dgrad fc2 dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU)
    case('dgrad fc1', 'NN', M, 4096, 1024, vendor=True)
    case('dgrad qkv', 'NN', M, 3072, 1024)
    case('dgrad proj', 'NN', M, 1024, 1024)
    case('wgrad fc1', 'TN', M, 4096, 1024, vendor=True)
    case('wgrad proj', 'TN', M, 1024, 1024)
    case('lm head (8w/4w forced)', 'NT', 8184, 50304, 1024, policy=2)
