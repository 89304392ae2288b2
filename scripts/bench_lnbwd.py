"""LayerNorm backward at the encoder shape (49512 x 1024), the two call forms of an encoder block.  PIXPARSE_AMD_LIB=<variant> python scripts/bench_lnbwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
hip.load()
dev = torch.device('cuda:0')
M, D = 49512, 1024
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, D, generator=g, device=dev)
dy16 = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
dy32 = torch.randn(M, D, generator=g, device=dev)
gamma = torch.rand(D, generator=g, device=dev) + 0.5
mean = x.mean(1); rstd = 1.0 / (x.var(1, unbiased=False) + 1e-5).sqrt()
dx32 = torch.zeros(M, D, device=dev); dx16 = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
dgamma = torch.zeros(D, device=dev); dbeta = torch.zeros(D, device=dev); cs = torch.zeros(D, device=dev)
forms = {'dy bf16 -> dx fp32 accumulate': lambda: ops.layernorm_bwd(None, dy16, x, gamma, mean, rstd, dx32, True, None, dgamma, dbeta),
         'dy fp32 + bf16 -> dx fp32 + bf16 + colsum': lambda: ops.layernorm_bwd(dy32, dy16, x, gamma, mean, rstd, dx32, False, dx16, dgamma, dbeta, dx_colsum=cs)}
for name, fn in forms.items():
    for rnd in range(2):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'{os.path.basename(os.environ.get("PIXPARSE_AMD_LIB", "product")):14s} {name:44s}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us', flush=True)
