#!/bin/bash
export PIXPARSE_AMD_SKIP_BUILD_CHECK=1   # objects are rebuilt by hand below, with other flags than build.py records
# A/B of attention.hip compile-time knobs on ONE box: rebuild the object with each -D set, relink, run the micro-bench twice.
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -c $C/attention.hip -o $C/attention.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== $flags"
  python scripts/bench_kernels.py attn 2>&1 | grep attn
  python scripts/bench_kernels.py attn 2>&1 | grep attn
done
