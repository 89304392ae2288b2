#!/bin/bash
# A/B of attention.hip compile-time knobs on ONE box: rebuild the object with each -D set, relink, run the micro-bench twice.
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $(extra_flags attention.hip) $flags -c $C/attention.hip -o $C/attention.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== $flags"
  python scripts/bench_kernels.py attn 2>&1 | grep attn
  python scripts/bench_kernels.py attn 2>&1 | grep attn
done
