#!/bin/bash
# correctness of a generator variant of the hand-placed attention backward (SPX_OPTS=...), then the default build is restored:  ab_spx_check.sh seedsvmem
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
SPX_OPTS="$1" python $C/gen_attn_bwd_sp.py || exit 1
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $(extra_flags attention.hip) -c $C/attention.hip -o $C/attention.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
python scripts/check_attn_sp.py --ref 2>&1 | grep "^OK\|^FAIL\|mode 2" | cut -c1-200
