#!/bin/bash
# alternates scripts/bench_attn_fwd.py (stream only) over the libraries given as arguments, three rounds
ulimit -c 0
for rep in 1 2 3; do for lib in "$@"; do echo "== $lib"; PIXPARSE_AMD_LIB=$lib timeout 120 python scripts/bench_attn_fwd.py one 2>&1 | grep -v amdgpu.ids | grep "1 wave"; done; done
