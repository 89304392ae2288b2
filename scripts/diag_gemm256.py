import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
torch.manual_seed(0)
for K in (64, 128, 192, 256, 320):
    M, N = 512, 512
    x = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.1).to(BF16)
    ref = x.float() @ w.float().t()
    hip.call('crl_gemm_set_policy', 2)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_fwd(x, w, None, out)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    bad = err > 0.05 + 0.02 * ref.abs()
    print('NT K', K, 'bad', int(bad.sum()), 'max', float(err.max()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('  bad rows', rows[:8].tolist(), '...', rows[-4:].tolist(), len(rows), ' bad cols', cols[:8].tolist(), '...', cols[-4:].tolist(), len(cols))
        # which K-tiles are missing? compare against partial sums
        for kt in range(K // 64):
            part = x[:, kt*64:(kt+1)*64].float() @ w[:, kt*64:(kt+1)*64].float().t()
            r = (ref - out.float())
            # projection of the residual on this tile's contribution
            coef = float((r * part).sum() / (part * part).sum())
            print('   residual coef on k-tile', kt, round(coef, 3))
    hip.call('crl_gemm_set_policy', 0)
