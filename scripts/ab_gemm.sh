#!/bin/bash
# A/B of gemm256.hip compile-time knobs on ONE box
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $(extra_flags gemm256.hip) $flags -c $C/gemm256.hip -o $C/gemm256.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== $flags"
  python scripts/bench_kernels.py gemm 2>&1 | grep -E "square 8192 .*pol=2|pol=0"
  python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gemm or wgrad" 2>&1 | tail -1
done
