#!/bin/bash
# Builds a variant of the WHOLE library with extra compile flags for every translation unit into pixparse_amd/csrc/variants/<name>.so
#   scripts/ab_full_variant.sh name "-DTILE_GROUP_M=1" [name2 "flags2" ...]
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
mkdir -p $C/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  D=/tmp/variant_$name; mkdir -p $D
  python - "$name" "$flags" <<'PY'
import sys, subprocess, os
from concurrent.futures import ThreadPoolExecutor
from pixparse_amd import build as b
name, flags = sys.argv[1], sys.argv[2].split()
D = f'/tmp/variant_{name}'
def one(src):
    o = os.path.join(D, os.path.splitext(src)[0] + '.o')
    cmd = [b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(src, []) + flags + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', os.path.join(b.CSRC, src), '-o', o]
    subprocess.run(cmd, check=True, capture_output=True)
    return o
with ThreadPoolExecutor(6) as ex:
    objs = list(ex.map(one, b.SOURCES))
subprocess.run([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(b.CSRC, 'variants', name + '.so')] + objs, check=True)
print('built', name, flags)
PY
done
