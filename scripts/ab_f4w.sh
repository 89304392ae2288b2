#!/bin/bash
# Builds variants of the library that differ in the generated forward-attention stream (F4W_OPTS / F4W_DROP / G4W_DROP environment of
# gen_attn_fwd4w.py) into pixparse_amd/csrc/variants/<name>.so (git-ignored, travels with gpurun) and restores the default stream.  Usage:
#   scripts/ab_f4w.sh name1 "F4W_DROP=exp" name2 "F4W_OPTS=bar=2" ...
# then on the GPU box: for v in ...; do PIXPARSE_AMD_LIB=pixparse_amd/csrc/variants/$v.so python scripts/bench_attn_fwd.py one; done
cd "$(dirname "$0")/.."
C=pixparse_amd/csrc
mkdir -p $C/variants
FL=$(python -c "from pixparse_amd import build; print(' '.join(build.FLAGS + build.EXTRA_FLAGS.get('attention.hip', [])))")
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  ( export $envs; cd $C && python gen_attn_fwd4w.py ) || exit 1
  hipcc $FL -DF4W_NO_FALLBACK=1 -DF4W_STAMPS=1 $F4W_CFLAGS -c $C/attention.hip -o /tmp/attention_$name.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$name.so /tmp/attention_$name.o $(ls $C/*.o | grep -v attention.o | tr "\n" " ") || exit 1
  echo "built $name ($envs)"
done
( cd $C && unset F4W_OPTS F4W_DROP G4W_DROP && python gen_attn_fwd4w.py )
