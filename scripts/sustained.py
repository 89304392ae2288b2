import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
hip.call('crl_gemm_set_policy', 2)
M = N = K = 8192
x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
out = torch.empty(M, N, dtype=BF16, device=dev)
ops.linear_fwd(x, w, None, out); torch.cuda.synchronize()
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4 if rep == 0 else 300
    e0.record()
    for _ in range(n): ops.linear_fwd(x, w, None, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'rep {rep}: n={n} {ms:.3f} ms/launch {2.0*M*N*K/ms/1e9:.0f} TF/s', flush=True)
os.system('rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6')
