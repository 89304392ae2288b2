#!/bin/bash
# same-box A/B of the whole train step: committed (HEAD) version vs working-tree version of ONE csrc file
#   gpurun -- 'bash scripts/ab_file.sh gemm.hip'       (run `git show HEAD:pixparse_amd/csrc/<file> > pixparse_amd/csrc/<file>.head` first: .git does not travel)
cd "$(dirname "$0")/.."
source scripts/_ab_common.sh
C=pixparse_amd/csrc
F=$1
OBJ=$C/${F%.*}.o
EXTRA="$(extra_flags $F)"
for v in head tree head tree; do
  if [ $v = head ]; then cp $C/$F.head $C/_ab_$F; else cp $C/$F $C/_ab_$F; fi
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA -c $C/_ab_$F -o $OBJ || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libcruller_hip.so $(ls $C/*.o | tr "\n" " ") || exit 1
  echo "== $v: $(python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "docs/s", d["ms_per_step"], "ms/step")')"
done
rm -f $C/_ab_$F
