#!/bin/bash
# round 6: bias gradients from the weight-gradient GEMM (column sums of A on the matrix pipe): the whole -m gpu suite, kernel timings with / without,
# the step with / without (PIXPARSE_AMD_FUSE_DBIAS), alternating
ulimit -c 0
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6_dbias_pytest.txt
cat gpurun_out/r6_dbias_pytest.txt
python - > gpurun_out/r6_dbias_kernels.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from pixparse_amd import hip, ops
hip.load(); dev = torch.device('cuda:0'); BF16 = torch.bfloat16
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in (('fc1 wgrad', 49512, 4096, 1024), ('qkv wgrad', 49512, 3072, 1024), ('dec fc1 wgrad', 8184, 4096, 1024), ('dec qkv wgrad', 8184, 3072, 1024), ('cross kv wgrad', 49512, 2048, 1024)):
    dy = torch.randn(M, N, device=dev).to(BF16); x = torch.randn(M, K, device=dev).to(BF16)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    for rep in range(2):
        a = timed(lambda: ops.linear_wgrad(dy, x, dw, True))
        b = timed(lambda: ops.linear_wgrad(dy, x, dw, True, dbias=db))
        c = timed(lambda: ops.colsum(dy, db, True))
        print(f'{name:16s} {M}x{N}x{K}: wgrad alone {a:7.1f} us | wgrad + fused bias gradient {b:7.1f} us | separate column-sum pass {c:6.1f} us | saved {a + c - b:6.1f} us', flush=True)
PY
cat gpurun_out/r6_dbias_kernels.txt
for rep in 1 2 3; do for f in 0 1; do
  echo "== PIXPARSE_AMD_FUSE_DBIAS=$f: $(PIXPARSE_AMD_FUSE_DBIAS=$f python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step loss", d["loss"])')"
done; done > gpurun_out/r6_dbias_step_ab.txt 2>&1
cat gpurun_out/r6_dbias_step_ab.txt
