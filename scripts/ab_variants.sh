#!/bin/bash
# Builds variants of the library with extra compile flags for ONE translation unit into pixparse_amd/csrc/variants/<name>.so.  Usage:
#   scripts/ab_variants.sh file.hip name1 "-DFLAG=1" name2 "-DOTHER=2" ...
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
mkdir -p $C/variants
F=$1; shift
FL=$(python -c "from pixparse_amd import build; print(' '.join(build.FLAGS + build.EXTRA_FLAGS.get('$F', [])))")
O=${F%.*}.o
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc $FL $flags -c $C/$F -o /tmp/${name}_$O || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$name.so /tmp/${name}_$O $(ls $C/*.o | grep -v "/$O" | tr "\n" " ") || exit 1
  echo "built $name ($F $flags)"
done
