#!/bin/bash
# same-box A/B of the WHOLE train step for gemm256.hip compile-time knobs: rebuild, relink, bench.py (no roofline / cpu legs)
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c $C/gemm256.hip -o $C/gemm256.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $C/gemm.o $C/gemm256.o $C/attention.o $C/rowops.o $C/loss_optim.o $C/swin.o $C/preprocess.o $C/skinny.o $C/attn_decode.o $C/capi.o || exit 1
  echo "== $flags: $(python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step")')"
done
