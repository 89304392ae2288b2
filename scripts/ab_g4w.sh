#!/bin/bash
# Builds variants of the library that differ in the generated gemm4w stream (G4W_OPTS / G4W_DROP environment of gen_gemm4w.py) into
# pixparse_amd/csrc/variants/<name>.so (git-ignored, travels with gpurun) and restores the default stream.  Usage:
#   scripts/ab_g4w.sh name1 "G4W_DROP=dma" name2 "G4W_OPTS=..." ...
# then on the GPU box: for v in ...; do PIXPARSE_AMD_LIB=pixparse_amd/csrc/variants/$v.so python scripts/bench_gemm4w.py one; done
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
mkdir -p $C/variants
FL=$(python -c "from pixparse_amd import build; print(' '.join(build.FLAGS + build.EXTRA_FLAGS.get('gemm4w.hip', [])))")
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  ( export $envs; cd $C && python gen_gemm4w.py ) || exit 1
  hipcc $FL -c $C/gemm4w.hip -o /tmp/gemm4w_$name.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$name.so /tmp/gemm4w_$name.o $(ls $C/*.o | grep -v gemm4w.o | tr "\n" " ") || exit 1
  echo "built $name ($envs)"
done
( cd $C && unset G4W_OPTS G4W_DROP && python gen_gemm4w.py )
