"""ctypes binding of libcruller_hip.so (include/crl.h) -- the only way compute reaches the GPU.

There is deliberately NO fallback: if the library is missing or a call fails this module raises.
The reference-side equivalent of this file is shown in INTEGRATION.md.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

# torch FIRST: it ships its own libamdhip64; if libcruller_hip.so were dlopen'ed before torch, /opt/rocm's copy would be
# mapped as well and the process would hold two HIP runtimes -- launches through the second one fail with
# "no ROCm-capable device is detected" once torch has initialised the GPU (seen with build() followed by smoke()).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libcruller_hip.so')

P, I, L, F, Z = c_void_p, c_int, c_int64, c_float, c_size_t
U32, U64 = c_uint32, c_uint64

# name -> (restype, argtypes); mirrors include/crl.h one to one
SIGNATURES = {
    'crl_version': (I, []),
    'crl_last_error': (c_char_p, []),
    'crl_gemm_ws_bytes': (Z, [I, I, L, L, L]),
    'crl_gemm_set_policy': (I, [I]),
    'crl_gemm_set_big_kernel': (I, [I]),
    'crl_gemm_set_overlap': (I, [I]),
    'crl_gemm_set_quant_cost': (I, [F]),
    'crl_gemm_calibrate_ws_bytes': (Z, []),
    'crl_gemm_calibrate': (I, [P, Z, P]),
    'crl_gemm_model': (I, [P, P, P]),
    'crl_gemm_set_schedule': (I, [I]),
    'crl_gemm_set_reserved_cus': (I, [I]),
    'crl_gemm_bf16': (I, [I, I, L, L, L, P, L, P, L, P, P, L, P, L, P, L, F, L, P, Z, P]),
    'crl_attn_decode_ws_bytes': (Z, [I, I, I]),
    'crl_attn_decode': (I, [P, L, P, L, L, P, L, L, P, L, I, I, I, F, P, P, L, P, Z, P]),
    'crl_linear_skinny_bf16': (I, [I, I, L, L, P, L, P, L, P, P, L, P, L, P, L, P]),
    'crl_linear_skinny_ln_bf16': (I, [I, I, L, L, P, L, P, P, F, P, L, P, L, P, P, L, P, L, P]),
    'crl_colsum_ws_bytes': (Z, [L]),
    'crl_colsum_bf16': (I, [P, L, L, L, P, I, P, P]),
    'crl_layernorm_fwd': (I, [P, P, P, F, L, L, P, P, P, P, P]),
    'crl_layernorm_bwd_ws_bytes': (Z, [L]),
    'crl_layernorm_bwd': (I, [P, P, P, P, P, P, L, L, P, I, P, P, P, P, I, P, P]),
    'crl_attn_fwd': (I, [P, L, L, P, L, L, P, L, L, P, L, L, P, I, I, I, I, F, I, I, F, U64, U32, U32, P]),
    'crl_attn_dropout_mask': (I, [P, I, I, I, I, F, U64, U32, U32, P]),
    'crl_attn_bwd': (I, [P, L, L, P, L, L, P, L, L, P, L, L, P, L, L, P, P, P, L, L, P, L, L, P, L, L,
                         I, I, I, I, F, I, I, F, U64, U32, U32, P, Z, P]),
    'crl_attn_bwd_ws_bytes': (Z, [I, I, I, I, I]),
    'crl_attn_fwd_set_mode': (I, [I]),
    'crl_attn_fwd_set_persistent': (I, [I]),
    'crl_attn_bwd_set_mode': (I, [I]),
    'crl_attn_bwd_set_parts': (I, [I]),
    'crl_attn_bwd_set_chain': (I, [I]),
    'crl_attn_bwd_set_qsplit': (I, [I]),
    'crl_attn_bwd_qsplit_for': (I, [I, I]),
    'crl_attn_bwd_set_persistent': (I, [I]),
    'crl_attn_bwd_chain_for': (I, [I, I]),
    'crl_debug_occupy_cus': (I, [I, ctypes.c_double, P, P]),
    'crl_prof_begin': (I, [I]),
    'crl_prof_end': (I, [I, P, P, P]),
    'crl_swin_attn_fwd': (I, [P, P, P, I, I, I, I, I, I, F, P]),
    'crl_swin_attn_bwd': (I, [P, P, P, P, P, I, I, I, I, I, I, F, P]),
    'crl_patch_merge_fwd': (I, [P, P, I, I, I, I, P]),
    'crl_patch_merge_bwd': (I, [P, P, I, I, I, I, P]),
    'crl_im2row': (I, [P, P, I, I, I, I, I, I, I, I, P]),
    'crl_vit_tokens_fwd': (I, [P, P, P, P, I, I, I, P]),
    'crl_vit_tokens_bwd': (I, [P, P, P, P, I, I, I, I, P]),
    'crl_embed_fwd': (I, [P, P, P, P, I, I, I, I, I, P]),
    'crl_embed_decode': (I, [P, P, P, P, I, I, I, I, P, P]),
    'crl_embed_bwd_ws_bytes': (Z, [I, I, I]),
    'crl_embed_bwd': (I, [P, P, P, P, I, I, I, I, I, I, P, Z, P]),
    'crl_cross_entropy': (I, [P, L, P, L, I, F, F, P, P, P, P, P, P]),
    'crl_grad_norm_ws_bytes': (Z, []),
    'crl_grad_norm': (I, [P, L, F, F, P, P, P]),
    'crl_grad_norm_scaled': (I, [P, L, F, F, F, F, I, P, P, P]),
    'crl_optim_prepare': (I, [P, F, F, F, I, I, F, F, P]),
    'crl_adamw': (I, [P, P, P, P, P, L, F, F, F, F, F, I, P, I, P]),
    'crl_cast_bf16': (I, [P, P, L, P]),
    'crl_cast_pad_bf16': (I, [P, P, L, L, L, P]),
    'crl_add_bf16_to_f32': (I, [P, P, L, I, P]),
    'crl_dropout': (I, [P, P, L, I, P, F, U64, U32, U32, P]),
    'crl_dropout_add': (I, [P, P, P, L, F, U64, U32, U32, P]),
    'crl_dropout_mask': (I, [P, L, F, U64, U32, U32, P]),
    'crl_droppath_scale': (I, [P, I, F, U64, U32, U32, P]),
    'crl_rowscale_add': (I, [P, P, P, P, L, L, L, P]),
    'crl_rowscale_bf16': (I, [P, P, P, L, L, L, P]),
    'crl_image_preprocess_u8': (I, [P, I, I, I, P, P, P, I, P, P, P, I, P, P, P, P, I, I, P]),
}

NT, NN, TN = 0, 1, 2
EPI_BF16, EPI_BF16_GELU, EPI_BF16_DGELU, EPI_F32_RESID, EPI_F32, EPI_F32_ACC = range(6)

_lib = None


class HipLibraryError(RuntimeError):
    pass


def _check_not_stale(path: str) -> None:
    """the library must have been built from the kernel sources lying next to it (build.py records their sha1 in build_info.json):
    a stale .so would silently run last week's kernels.  scripts that rebuild objects by hand (scripts/ab_*.sh) set
    PIXPARSE_AMD_SKIP_BUILD_CHECK=1."""
    import json
    from . import build as _build
    if not os.path.exists(_build.BUILD_INFO):
        return                                  # built by hand (explicit hipcc line): nothing recorded, nothing to compare
    with open(_build.BUILD_INFO) as f:
        built = json.load(f).get('sources', {})
    now = _build.source_digest()
    changed = sorted(k for k in now if built.get(k) != now[k])
    if changed:
        raise HipLibraryError(f'{path} is older than its sources ({", ".join(changed)} changed since it was built): run '
                              '`python -m pixparse_amd.build` (or __graft_entry__.build()); set PIXPARSE_AMD_SKIP_BUILD_CHECK=1 to load it anyway')


def load(path: str = None) -> ctypes.CDLL:
    """dlopen the library and bind every symbol crl.h declares; raises if any is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get('PIXPARSE_AMD_LIB') or LIB_PATH      # PIXPARSE_AMD_LIB: same-box A/B against another build of the library
    if not os.path.exists(path):
        raise HipLibraryError(
            f'{path} not found: build it with `python -m pixparse_amd.build` (hipcc --offload-arch=gfx950). '
            'pixparse_amd has no CPU or PyTorch fallback path.')
    if path == LIB_PATH and os.environ.get('PIXPARSE_AMD_SKIP_BUILD_CHECK', '0') != '1':
        _check_not_stale(path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if os.environ.get('PIXPARSE_AMD_LIB') == path:
                continue                        # an older build under A/B may lack newer entry points: using one raises AttributeError
            raise HipLibraryError(f'{path} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().crl_last_error().decode()


def call(name: str, *args) -> None:
    """invoke an int-returning entry point; non-zero return -> HipLibraryError(crl_last_error())."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise HipLibraryError(f'{name} failed ({rc}): {last_error()}')


def query(name: str, *args) -> int:
    return int(getattr(load(), name)(*args))
