import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
hip.call('crl_gemm_set_policy', 2)
M = N = K = 8192
x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
out = torch.empty(M, N, dtype=BF16, device=dev)
for _ in range(4): ops.linear_fwd(x, w, None, out)
torch.cuda.synchronize()
