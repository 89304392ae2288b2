"""Per-tile timeline of the persistent 256x256 GEMM kernel (debug build: gemm256.hip compiled with -DG_TIMING=<workgroup>).
    hipcc ... -DG_TIMING=0 -c pixparse_amd/csrc/gemm256.hip ; relink ; python scripts/gemm_timeline.py
Prints, per tile iteration of that workgroup's wave 0, the s_memtime deltas (in clocks and us at 100 MHz-independent shader clock):
  wait   tile top -> prefetched operands landed + previous stores acknowledged + barrier
  loop   K loop
  drain  end of loop -> outstanding DMA drained
  issue  next tile's first six half-tiles issued
  epi    epilogue (until its last store is issued)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
lib = hip.load()
lib.crl_gemm_debug_read.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 128)()
def run(name, layout, M, N, K, epi):
    if layout == 'NT':
        x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16 if epi in (0, 1, 2) else torch.float32, device=dev)
        aux = torch.empty(M, N, dtype=BF16, device=dev) if epi == ops.EPI_BF16_GELU else None
        fn = lambda: ops.linear_fwd(x, w, torch.randn(N, device=dev), out, epi, aux=aux, resid=out if epi == ops.EPI_F32_RESID else None)
    else:
        dy = torch.randn(M, N, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, K, dtype=BF16, device=dev)
        aux = torch.randn(M, K, device=dev).to(BF16) if epi == ops.EPI_BF16_DGELU else None
        fn = lambda: ops.linear_dgrad(dy, w, out, epi, aux=aux)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert lib.crl_gemm_debug_read(buf) == 0
    t = [[buf[i * 8 + j] for j in range(8)] for i in range(16)]
    print(f'== {name} {layout} M={M} N={N} K={K} epi={epi}: {e0.elapsed_time(e1) * 1e3:.1f} us (100 clocks = 1 us at 100 MHz; shader clock ~ 2 GHz => /2000)')
    prev5 = None
    for i in range(16):
        r = t[i]
        if r[5] == 0 or (i and r[0] < t[i - 1][0]): break
        d = lambda a, b: (r[b] - r[a])
        gap = (r[0] - prev5) if prev5 else 0
        print(f'  tile {i}: zero+{gap:6d} wait {d(0,1):6d} loop {d(1,2):7d} drain {d(2,3):6d} issue {d(3,4):6d} epi {d(4,5):7d}  total {(r[5]-r[0]) + gap:7d}')
        prev5 = r[5]
M = int(os.environ.get('TL_M', 49512))
run('qkv', 'NT', M, 3072, 1024, 0)
run('proj resid', 'NT', M, 1024, 1024, ops.EPI_F32_RESID)
run('fc1 gelu', 'NT', M, 4096, 1024, ops.EPI_BF16_GELU)
run('dgrad dgelu', 'NN', M, 1024, 4096, ops.EPI_BF16_DGELU)
run('dgrad proj', 'NN', M, 1024, 1024, 0)
