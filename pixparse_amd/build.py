"""Builds pixparse_amd/csrc/libcruller_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m pixparse_amd.build            # incremental
    python -m pixparse_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcruller_hip.so')
SOURCES = ['gemm.hip', 'gemm256.hip', 'gemm4w.hip', 'attention.hip', 'rowops.hip', 'loss_optim.hip', 'swin.hip', 'preprocess.hip', 'skinny.hip', 'attn_decode.hip', 'dropout.hip', 'debug.hip', 'capi.cpp']
HEADERS = ["common.h", "gemm_common.h", "gemm_epilogue.h", "attn_frag.h", "attn_bwd_sp_body.inc", "attn_fwd4w_body.inc", "attn_fwd2x_body.inc", "gemm4w_body_nt.inc", "gemm4w_body_nn.inc", "gemm4w_body_tn.inc", "gemm4w_body_tn_cs.inc", "gemm4w_body_nt_ovl.inc", "gemm4w_body_nn_ovl.inc", "gemm4w_drain_ovl.inc", os.path.join('..', '..', 'include', 'crl.h')]
# generated sources: (generator script, output) -- the output is committed; it is regenerated when the script is newer
GENERATED = [('gen_attn_bwd_sp.py', 'attn_bwd_sp_body.inc'), ('gen_attn_fwd4w.py', 'attn_fwd4w_body.inc'), ('gen_gemm4w.py', 'gemm4w_body_nt.inc')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result']
# per-file extras: hipcc's SLP vectoriser packs the softmax / dS multiplies of the attention kernels into v_pk_mul_f32 on
# misaligned register pairs and then spends ~25 v_mov / v_perm / v_alignbit per 32x32 block re-assembling the bf16 MFMA
# operand; these kernels are VALU-issue bound, so it is switched off for them.
# the persistent GEMMs pull their next tile with ONE lane's returning atomic whose answer is consumed a whole K loop later; LLVM's
# atomic optimizer would rewrite it into a wave-aggregated add + readfirstlane and wait for it on the spot.
_NO_ATOMIC_OPT = ['-mllvm', '-amdgpu-atomic-optimizer-strategy=None']
EXTRA_FLAGS = {'attention.hip': ['-fno-slp-vectorize'], 'gemm256.hip': _NO_ATOMIC_OPT, 'gemm4w.hip': _NO_ATOMIC_OPT}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(out, deps):
    """mtime rule -- only for the GENERATED sources (generator newer than its output) and the link step (an object newer than the library);
    whether an OBJECT is reused is decided by content digests (below)"""
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _sha1(path):
    import hashlib
    return hashlib.sha1(open(path, 'rb').read()).hexdigest()


def _unit_key(src, hipcc):
    """what an object file is a function of: the translation unit, every header / generated stream of the library (any of them may be
    included), the compiler and the flags.  An object is reused only when the key recorded at ITS compilation equals today's key."""
    import hashlib
    h = hashlib.sha1()
    for f in [src] + sorted(HEADERS):
        h.update(f.encode() + b'\0' + _sha1(os.path.join(CSRC, f)).encode() + b'\0')
    h.update(' '.join([hipcc] + FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()


BUILD_INFO = os.path.join(CSRC, 'build_info.json')      # written next to the library; git-ignored like the .so, travels with it


def source_digest() -> dict:
    """sha1 of every kernel source and header the library is built from"""
    return {f: _sha1(os.path.join(CSRC, f)) for f in sorted(SOURCES + HEADERS)}


def _read_info() -> dict:
    import json
    try:
        with open(BUILD_INFO) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def generate(force: bool = False, verbose: bool = True) -> None:
    for script, out in GENERATED:
        sp, op = os.path.join(CSRC, script), os.path.join(CSRC, out)
        if force or _stale(op, [sp]):
            if verbose:
                print('[build] generating', out, flush=True)
            subprocess.run([sys.executable, sp], check=True)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    generate(force, verbose)
    prev = _read_info()
    prev_keys = prev.get('object_keys', {})
    keys = {src: _unit_key(src, hipcc) for src in SOURCES}
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        # reuse an object only if it was compiled from exactly these bytes with exactly these flags (an mtime rule
        # would bless an object that is newer than a source whose content went back)
        if force or not os.path.exists(o) or prev_keys.get(src) != keys[src]:
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', s, '-o', o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print('[build]', ' '.join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), r.stderr))

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    linked = bool(force or jobs or _stale(LIB, objs) or prev.get('linked_keys') != keys)
    if linked:
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    # what this call did, and the sources the library now corresponds to: hip.load() refuses a library whose sources have changed since
    import json
    import time
    compiled = [os.path.basename(c[-3]) for c in jobs]
    info = dict(time=time.strftime('%Y-%m-%d %H:%M:%S'), hipcc=hipcc, flags=FLAGS, extra_flags=EXTRA_FLAGS,
                compiled=compiled, reused=[s for s in SOURCES if s not in compiled], linked=linked, sources=source_digest(),
                object_keys=keys, linked_keys=keys)
    with open(BUILD_INFO, 'w') as f:
        json.dump(info, f, indent=1)
    if verbose:
        print(f'[build] compiled {len(compiled)} of {len(SOURCES)} translation units ({", ".join(compiled) or "all objects up to date"}); '
              f'linked: {linked}', flush=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
