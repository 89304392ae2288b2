#!/bin/bash
# same-box A/B of the whole train step for compile-time knobs of ONE csrc file:  ab_flags.sh attention.hip "-DFWD_OCC=4" "-DFWD_OCC=3" ...
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
F=$1; shift
OBJ=$C/${F%.*}.o
EXTRA="$(extra_flags $F)"
for flags in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA $flags -c $C/$F -o $OBJ || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== $flags: $(python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step")')"
done
