#!/bin/bash
# same-box A/B of the whole cfg-3 train step between BUILT libraries, alternating, three rounds:
#   ab_libs_step.sh pixparse_amd/csrc/variants/libcruller_a.so pixparse_amd/csrc/libcruller_hip.so ...
# (PIXPARSE_AMD_LIB loads the given library instead of the in-tree build; the Python side is the working tree's for every arm)
ulimit -c 0
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
for rep in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    echo "== $lib: $(PIXPARSE_AMD_LIB=$lib PIXPARSE_AMD_SKIP_BUILD_CHECK=1 python bench.py --no-cpu-baseline --no-roofline --no-host-leg --no-peak --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step loss", d["loss"])')"
  done
done
