#!/bin/bash
# same-box timing of generator variants of the hand-placed attention backward (SPX_DROP classes: WRONG RESULTS, timing only), then the
# default build is restored.   bash scripts/ab_spx.sh "" exp "exp,mul,cvt" ...
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
build() {
  SPX_DROP="$1" SPX_OPTS="$2" python $C/gen_attn_bwd_sp.py || exit 1
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $(extra_flags attention.hip) $SPX_CFLAGS -c $C/attention.hip -o $C/attention.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
}
for v in "$@"; do
  drop="${v%%:*}"; opts=""; [[ "$v" == *:* ]] && opts="${v#*:}"
  build "$drop" "$opts"
  echo "== drop=[$drop] opts=[$opts]"
  python scripts/check_attn_sp.py --time-only --modes 2 2>&1 | grep "mode 2\|stamps\|trace\|prologue"
done
