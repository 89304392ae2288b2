#!/bin/bash
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
for v in old new old new; do
  if [ $v = old ]; then cp $C/gemm256_old.hip.txt /tmp/g.hip; else cp $C/gemm256.hip /tmp/g.hip; fi
  cp /tmp/g.hip $C/_ab_gemm256.hip
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c $C/_ab_gemm256.hip -o $C/gemm256.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $C/gemm.o $C/gemm256.o $C/attention.o $C/rowops.o $C/loss_optim.o $C/swin.o $C/preprocess.o $C/skinny.o $C/attn_decode.o $C/capi.o || exit 1
  echo "== $v"
  python scripts/bench_kernels.py gemm 2>&1 | grep -E "pol=0" | grep -E "proj resid|fc1 gelu|fc1 plain|fc2 resid|dgelu|qkv"
  echo "   step: $(python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step")')"
done
rm -f $C/_ab_gemm256.hip
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gemm or wgrad" 2>&1 | tail -1
