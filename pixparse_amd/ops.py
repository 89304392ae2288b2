"""Tensor-level wrappers over the C-ABI (pixparse_amd.hip).  PyTorch is only the owner of device
memory and streams here: every function takes torch tensors, passes raw pointers + the current
HIP stream, and returns nothing that was computed by torch."""
from typing import Optional

import torch

from . import hip
from .hip import NT, NN, TN, EPI_BF16, EPI_BF16_GELU, EPI_BF16_DGELU, EPI_F32_RESID, EPI_F32, EPI_F32_ACC  # noqa: F401

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32
_FUSE_DBIAS = __import__('os').environ.get('PIXPARSE_AMD_FUSE_DBIAS', '1') != '0'     # A/B switch: bias gradients from the weight-gradient GEMM (default) or a pass of their own
VOCAB_PAD = 128   # logits row stride / embedding rows are padded to a multiple of this
K_PAD = 64        # contraction dims of NT/NN GEMMs are padded to a multiple of this when needed


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f'{name}: expected a cuda {dtype} tensor, got {t.dtype} on {t.device}')


def gemm(layout: int, epi: int, M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int,
         C: torch.Tensor, ldc: int, bias: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None,
         ldaux: int = 0, resid: Optional[torch.Tensor] = None, ldr: int = 0, colscale: float = 1.0, colscale_cols: int = 0) -> None:
    _chk(A, BF16, 'gemm A')
    _chk(B, BF16, 'gemm B')
    ws = None
    _gemm_env()
    wsb = hip.query('crl_gemm_ws_bytes', layout, epi, M, N, K)   # wgrad split-K slabs, or the split remainder rows of a forward / dgrad GEMM
    if wsb:
        ws = _gemm_scratch.get(wsb, A.device)
    hip.call('crl_gemm_bf16', layout, epi, M, N, K, _p(A), lda, _p(B), ldb, _p(bias), _p(C), ldc, _p(aux), ldaux,
             _p(resid), ldr, float(colscale), int(colscale_cols), _p(ws), wsb, _stream())


_gemm_env_applied = False


def _gemm_env():
    """A/B switches of the GEMM dispatch, read once: PIXPARSE_AMD_GEMM_BIG=0|1|2 (crl_gemm_set_big_kernel: 8-wave / 4-wave / per launch),
    PIXPARSE_AMD_GEMM_OVERLAP=0|1 (crl_gemm_set_overlap)"""
    global _gemm_env_applied
    if _gemm_env_applied:
        return
    _gemm_env_applied = True
    import os
    for env, fn in (('PIXPARSE_AMD_GEMM_BIG', 'crl_gemm_set_big_kernel'), ('PIXPARSE_AMD_GEMM_OVERLAP', 'crl_gemm_set_overlap')):
        if os.environ.get(env):
            hip.call(fn, int(os.environ[env]))


_calibrated = {}


def gemm_calibrate(device, force: bool = False):
    """one-off per process and device: refit the wave-quantisation cost model of the persistent GEMMs to what THIS device sustains
    (crl_gemm_calibrate; synchronises).  OPT-IN since round 5 (PIXPARSE_AMD_GEMM_CALIBRATE=1, or force=True): the fitted constants decide
    where the wave-quantisation cut falls, i.e. which rows take the split-contraction path with its different summation order, so a default-on
    timing fit made the bits of every large GEMM depend on measurement noise (run to run, rank to rank, across a resume) for no
    measured gain (244.3 vs 244.0 ms).  Returns (a_us, b_us, calibrated)."""
    import ctypes
    import os
    key = str(device)
    # (two ranks sharing one device -- CRL_DEBUG_SHARED_GPU, a validation aid -- would time each other's launches: built-in constants there)
    if (force or (key not in _calibrated and os.environ.get('PIXPARSE_AMD_GEMM_CALIBRATE', '0') == '1')) and os.environ.get('CRL_DEBUG_SHARED_GPU', '0') != '1':
        nbytes = hip.query('crl_gemm_calibrate_ws_bytes')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)      # its own buffer: released again right away
        rc = hip.load().crl_gemm_calibrate(_p(ws), nbytes, _stream())
        if rc < 0:
            raise hip.HipLibraryError(f'crl_gemm_calibrate failed ({rc}): {hip.last_error()}')
        del ws
        _calibrated[key] = rc
    a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
    hip.call('crl_gemm_model', ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c))
    return a.value, b.value, bool(c.value)


def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi: int = EPI_BF16,
               aux: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, n: Optional[int] = None,
               colscale: float = 1.0, colscale_cols: int = 0) -> None:
    """out[M, N] = x[M, K] @ w[N, K]^T (+bias) with the epilogue; x, w bf16 row-major 2-D (may be row-strided).
    colscale / colscale_cols: the first colscale_cols output columns are multiplied by colscale before the bf16 rounding (q of q|k|v)."""
    M, K = x.shape
    N = n if n is not None else w.shape[0]
    gemm(NT, epi, M, N, K, x, x.stride(0), w, w.stride(0), out, out.stride(0), bias, aux,
         aux.stride(0) if aux is not None else 0, resid, resid.stride(0) if resid is not None else 0, colscale, colscale_cols)


SKINNY_MAX_ROWS = 16


def linear_skinny(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi: int = EPI_BF16,
                  resid: Optional[torch.Tensor] = None, n: Optional[int] = None, out_row: Optional[torch.Tensor] = None,
                  out_row_stride: int = 0) -> None:
    """decode-time linear_fwd for M <= 16 rows (crl_linear_skinny_bf16): every weight byte read once, no tiles.
    out_row (device int32 scalar): the output is shifted by out_row * out_row_stride elements (KV-cache row of the step)."""
    M, K = x.shape
    N = n if n is not None else w.shape[0]
    hip.call('crl_linear_skinny_bf16', epi, M, N, K, _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0),
             _p(resid), resid.stride(0) if resid is not None else 0, _p(out_row), int(out_row_stride), _stream())


def linear_skinny_ln(x32: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, h32: Optional[torch.Tensor], w: torch.Tensor,
                     bias: Optional[torch.Tensor], out: torch.Tensor, epi: int = EPI_BF16, n: Optional[int] = None,
                     out_row: Optional[torch.Tensor] = None, out_row_stride: int = 0) -> None:
    """out = epilogue(bf16(LayerNorm(x32)) @ w^T + bias) for M <= 16 rows, h32 (optional) = the fp32 LayerNorm output
    (crl_linear_skinny_ln_bf16: the LayerNorm in front of a decode-step projection without a launch of its own)"""
    M, K = x32.shape
    N = n if n is not None else w.shape[0]
    hip.call('crl_linear_skinny_ln_bf16', epi, M, N, K, _p(x32), x32.stride(0), _p(gamma), _p(beta), float(eps), _p(h32),
             h32.stride(0) if h32 is not None else 0, _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), _p(out_row), int(out_row_stride), _stream())


def linear_dgrad(dy: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epi: int = EPI_BF16,
                 aux: Optional[torch.Tensor] = None, k: Optional[int] = None) -> None:
    """out[M, Kin] = dy[M, N] @ w[N, Kin]; contraction over N (= dy.shape[1])."""
    M, N = dy.shape
    Kin = k if k is not None else w.shape[1]
    gemm(NN, epi, M, Kin, N, dy, dy.stride(0), w, w.stride(0), out, out.stride(0), None, aux,
         aux.stride(0) if aux is not None else 0)


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool = True,
                 n: Optional[int] = None, k: Optional[int] = None, dbias: Optional[torch.Tensor] = None, dbias_accumulate: bool = True) -> None:
    """dw[N, Kin] (+)= dy[M, N]^T @ x[M, Kin]  (fp32 output straight into the grad arena).
    dbias (optional, fp32 [N]): the bias gradient dbias[n] (+)= sum_m dy[m, n] from the SAME call -- the 4-wave weight-gradient kernel sums the
    columns of its A operand on the matrix pipe while it streams them (crl_gemm_bf16, CRL_TN with aux); any other kernel: the column-sum pass."""
    M = dy.shape[0]
    N = n if n is not None else dy.shape[1]
    Kin = k if k is not None else x.shape[1]
    if dbias is not None:
        _chk(dbias, F32, 'linear_wgrad dbias')
        if not _FUSE_DBIAS:       # A/B switch (PIXPARSE_AMD_FUSE_DBIAS=0): the separate column-sum pass of rounds 1-5
            gemm(TN, EPI_F32_ACC if accumulate else EPI_F32, N, Kin, M, dy, dy.stride(0), x, x.stride(0), dw, dw.stride(0))
            colsum(dy, dbias, dbias_accumulate, n=n)
            return
    gemm(TN, EPI_F32_ACC if accumulate else EPI_F32, N, Kin, M, dy, dy.stride(0), x, x.stride(0), dw, dw.stride(0),
         aux=dbias, ldaux=1 if (dbias is not None and dbias_accumulate) else 0)


class Scratch:
    """lazily grown device scratch shared by the reductions (never read across calls).  Once a hipGraph has been captured
    (freeze_scratch()) a buffer that has to grow is RETIRED, not freed: the graph's launches keep writing their slabs / partials into
    the address they were captured with, which stays owned by this object instead of going back to the caching allocator where other
    tensors would be handed the same memory; from then on it grows by at least half its size."""
    frozen = False

    def __init__(self):
        self.buf = None
        self.retired = []

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            want = max(nbytes, 1 << 20)
            if Scratch.frozen and self.buf is not None:
                self.retired.append(self.buf)
                # a retired buffer is never released: grow geometrically so that a rising sequence of requests (eval / generation shapes after a
                # graphed training run) retires at most ~2 x the final size in all, not one buffer per request
                if self.buf.device == device:
                    want = max(want, self.buf.numel() + self.buf.numel() // 2)
            self.buf = torch.empty(want, dtype=torch.uint8, device=device)
            import os
            if os.environ.get('PIXPARSE_AMD_POISON', '0') == '1':    # debug aid (see layers/engines.py Buffers)
                self.buf.fill_(255)
        return self.buf


def freeze_scratch() -> None:
    """call before capturing launches into a hipGraph: from now on scratch buffers are never released (see Scratch)"""
    Scratch.frozen = True


_scratch = Scratch()
_gemm_scratch = Scratch()
_attn_scratch = Scratch()


def colsum(x: torch.Tensor, out: torch.Tensor, accumulate: bool = True, n: Optional[int] = None) -> None:
    M = x.shape[0]
    N = n if n is not None else x.shape[1]
    ws = _scratch.get(hip.query('crl_colsum_ws_bytes', N), x.device)
    hip.call('crl_colsum_bf16', _p(x), M, N, x.stride(0), _p(out), int(accumulate), _p(ws), _stream())


def layernorm_fwd(x, gamma, beta, eps, y_f32, y_bf16, mean, rstd) -> None:
    M, D = x.shape
    _chk(x, F32, 'layernorm x')
    hip.call('crl_layernorm_fwd', _p(x), _p(gamma), _p(beta), float(eps), M, D, _p(y_f32), _p(y_bf16), _p(mean), _p(rstd),
             _stream())


def layernorm_bwd(dy_f32, dy_bf16, x, gamma, mean, rstd, dx_f32, dx_accumulate, dx_bf16, dgamma, dbeta,
                  acc_wgrad: bool = True, dx_colsum: Optional[torch.Tensor] = None) -> None:
    """dx_colsum [D] (+)= column sums of the bf16 gradient written to dx_bf16 (the bias gradient of the Linear it feeds)"""
    M, D = x.shape
    assert dx_colsum is None or dx_bf16 is not None
    ws = _scratch.get(hip.query('crl_layernorm_bwd_ws_bytes', D), x.device)
    hip.call('crl_layernorm_bwd', _p(dy_f32), _p(dy_bf16), _p(x), _p(gamma), _p(mean), _p(rstd), M, D, _p(dx_f32),
             int(dx_accumulate), _p(dx_bf16), _p(dgamma), _p(dbeta), _p(dx_colsum), int(acc_wgrad), _p(ws), _stream())


HEAD_DIM = 64   # the attention kernels index head h at channel h * 64


def _chk_heads(heads: int, *tensors) -> None:
    for t in tensors:
        if t.shape[-1] != heads * HEAD_DIM:
            raise ValueError(f'attention kernels are built for head_dim {HEAD_DIM}: got {t.shape[-1]} channels for {heads} heads '
                             f'(head_dim {t.shape[-1] / max(heads, 1):g})')


def _bs_rs(t: torch.Tensor):
    """(batch stride, row stride) in elements of a [B, N, H*64-ish] view whose last dim is contiguous."""
    assert t.dim() == 3 and t.stride(2) == 1, 'attention operands are [B, N, H*64] views with contiguous channels'
    return t.stride(0), t.stride(1)


def _drop_args(drop, site):
    """(p, seed, step, site) of an attention-probability dropout call; drop: DropSpec-like with .p_attn (None / 0 = off)"""
    if drop is None or not getattr(drop, 'p_attn', 0.0):
        return 0.0, 0, 0, 0
    return float(drop.p_attn), drop.seed, drop.step, int(site)


LOG2E = 1.4426950408889634


_attn_fwd_env_applied = False


def attn_fwd(q, k, v, o, lse, heads: int, scale: float, causal: bool, drop=None, site: int = 0, q_prescaled: bool = False) -> None:
    """q [B,Nq,H*64], k/v [B,Nk,H*64] (strided views allowed), o like q, lse [B,H,Nq] f32.
    drop (DropSpec with p_attn > 0) + site: dropout of the attention probabilities (the backward call must pass the same pair).
    q_prescaled: q already carries scale * log2(e) (linear_fwd(..., colscale=scale * LOG2E, colscale_cols=D)): the faster forward kernel."""
    global _attn_fwd_env_applied
    if not _attn_fwd_env_applied:      # A/B switch: PIXPARSE_AMD_ATTN_FWD_MODE=1 the compiler-scheduled kernels, 3 / 4 the stream at one / two workgroups per CU
        import os
        _attn_fwd_env_applied = True
        if os.environ.get('PIXPARSE_AMD_ATTN_FWD_MODE'):
            hip.call('crl_attn_fwd_set_mode', int(os.environ['PIXPARSE_AMD_ATTN_FWD_MODE']))
        if os.environ.get('PIXPARSE_AMD_ATTN_FWD_PERSIST'):     # 0: one workgroup per query block instead of the persistent launch
            hip.call('crl_attn_fwd_set_persistent', int(os.environ['PIXPARSE_AMD_ATTN_FWD_PERSIST']))
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    _chk_heads(heads, q, k, v, o)
    hip.call('crl_attn_fwd', _p(q), *_bs_rs(q), _p(k), *_bs_rs(k), _p(v), *_bs_rs(v), _p(o), *_bs_rs(o), _p(lse),
             B, heads, Nq, Nk, float(scale), int(causal), int(q_prescaled), *_drop_args(drop, site), _stream())


def attn_dropout_mask(B: int, H: int, Nq: int, Nk: int, drop, site: int, device) -> torch.Tensor:
    """the keep mask [B, H, Nq, Nk] (uint8) the attention kernels regenerate for (drop, site)"""
    keep = torch.empty(B, H, Nq, Nk, dtype=torch.uint8, device=device)
    p, seed, step, st = _drop_args(drop, site)
    hip.call('crl_attn_dropout_mask', _p(keep), B, H, Nq, Nk, p, seed, step, st, _stream())
    return keep


def attn_decode(q, k, v, o, heads: int, scale: float, nk_minus1: Optional[torch.Tensor] = None,
                q_row: Optional[torch.Tensor] = None, q_row_stride: int = 0) -> None:
    """single-query attention over a KV cache: q, o [B, H*64] (row-strided), k / v [B, Nk, H*64] strided views.
    nk_minus1 (device int32 scalar): only the first nk_minus1 + 1 keys are valid (k.shape[1] is the capacity);
    q_row: q is taken q_row * q_row_stride elements further (the step's cache row)."""
    B = q.shape[0]
    Nk = k.shape[1]
    assert q.dim() == 2 and q.stride(1) == 1 and o.stride(1) == 1
    _chk_heads(heads, q, k, v, o)
    nbytes = hip.query('crl_attn_decode_ws_bytes', B, heads, Nk)
    ws = _scratch.get(nbytes, q.device)
    hip.call('crl_attn_decode', _p(q), q.stride(0), _p(k), *_bs_rs(k), _p(v), *_bs_rs(v), _p(o), o.stride(0), B, heads, Nk,
             float(scale), _p(nk_minus1), _p(q_row), int(q_row_stride), _p(ws), nbytes, _stream())


def embed_decode(ids, tok, pos, out, step: torch.Tensor, pos_offset: int = 2) -> None:
    """generation: ids [B, 1] at position `step` (device int32 scalar) -> out f32 [B, D]"""
    hip.call('crl_embed_decode', _p(ids), _p(tok), _p(pos), _p(out), ids.shape[0], out.shape[1], pos_offset, tok.shape[0], _p(step),
             _stream())


_attn_mode_env_applied = False


def attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, heads: int, scale: float, causal: bool, drop=None, site: int = 0,
             q_prescaled: bool = False) -> None:
    global _attn_mode_env_applied
    if not _attn_mode_env_applied:      # A/B switch: PIXPARSE_AMD_ATTN_BWD_MODE=1 forces the two-pass backward, 2 the single pass (crl_attn_bwd_set_mode)
        import os
        _attn_mode_env_applied = True
        if os.environ.get('PIXPARSE_AMD_ATTN_BWD_MODE'):
            hip.call('crl_attn_bwd_set_mode', int(os.environ['PIXPARSE_AMD_ATTN_BWD_MODE']))
        if os.environ.get('PIXPARSE_AMD_ATTN_BWD_QSPLIT'):  # query split of the single pass's remainder chains (crl_attn_bwd_set_qsplit; -1 = auto)
            hip.call('crl_attn_bwd_set_qsplit', int(os.environ['PIXPARSE_AMD_ATTN_BWD_QSPLIT']))
        if os.environ.get('PIXPARSE_AMD_ATTN_BWD_PERSIST'):  # 0: one workgroup per chain instead of the persistent ticket-pulling launch (crl_attn_bwd_set_persistent)
            hip.call('crl_attn_bwd_set_persistent', int(os.environ['PIXPARSE_AMD_ATTN_BWD_PERSIST']))
        if os.environ.get('PIXPARSE_AMD_ATTN_BWD_CHAIN'):   # key blocks per workgroup of the single pass (crl_attn_bwd_set_chain; 0 = auto)
            hip.call('crl_attn_bwd_set_chain', int(os.environ['PIXPARSE_AMD_ATTN_BWD_CHAIN']))
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    _chk_heads(heads, q, k, v, o, d_o, dq, dk, dv)
    wsb = hip.query('crl_attn_bwd_ws_bytes', B, heads, Nq, Nk, int(causal))     # partial-dQ slabs of the single-pass form (0: two-pass)
    ws = _attn_scratch.get(wsb, q.device) if wsb else None
    hip.call('crl_attn_bwd', _p(q), *_bs_rs(q), _p(k), *_bs_rs(k), _p(v), *_bs_rs(v), _p(o), *_bs_rs(o),
             _p(d_o), *_bs_rs(d_o), _p(lse), _p(delta), _p(dq), *_bs_rs(dq), _p(dk), *_bs_rs(dk), _p(dv), *_bs_rs(dv),
             B, heads, Nq, Nk, float(scale), int(causal), int(q_prescaled), *_drop_args(drop, site), _p(ws), wsb, _stream())


def swin_attn_fwd(qkv, table, out, B, Hf, Wf, heads, w, shift, scale) -> None:
    hip.call('crl_swin_attn_fwd', _p(qkv), _p(table), _p(out), B, Hf, Wf, heads, w, shift, float(scale), _stream())


def swin_attn_bwd(qkv, table, d_out, dqkv, dtable, B, Hf, Wf, heads, w, shift, scale) -> None:
    hip.call('crl_swin_attn_bwd', _p(qkv), _p(table), _p(d_out), _p(dqkv), _p(dtable), B, Hf, Wf, heads, w, shift,
             float(scale), _stream())


def patch_merge_fwd(x, y, B, Hf, Wf, C) -> None:
    hip.call('crl_patch_merge_fwd', _p(x), _p(y), B, Hf, Wf, C, _stream())


def patch_merge_bwd(dy, dx, B, Hf, Wf, C) -> None:
    hip.call('crl_patch_merge_bwd', _p(dy), _p(dx), B, Hf, Wf, C, _stream())


def im2row(image, patches, P, gh, gw) -> None:
    B, C, H, W = image.shape
    _chk(image, F32, 'im2row image')
    hip.call('crl_im2row', _p(image), _p(patches), B, C, H, W, P, gh, gw, patches.shape[1], _stream())


def vit_tokens_fwd(patch, cls, pos, x, B, Np, D) -> None:
    hip.call('crl_vit_tokens_fwd', _p(patch), _p(cls), _p(pos), _p(x), B, Np, D, _stream())


def vit_tokens_bwd(dx, dpatch, dcls, dpos, B, Np, D, accumulate: bool = True) -> None:
    hip.call('crl_vit_tokens_bwd', _p(dx), _p(dpatch), _p(dcls), _p(dpos), int(accumulate), B, Np, D, _stream())


def embed_fwd(ids, tok, pos, out, pos_offset: int = 2) -> None:
    B, T = ids.shape
    hip.call('crl_embed_fwd', _p(ids), _p(tok), _p(pos), _p(out), B, T, out.shape[1], pos_offset, tok.shape[0], _stream())


def embed_bwd(ids, dt, dtok, dpos, pos_offset: int = 2, accumulate: bool = True) -> None:
    B, T = ids.shape
    nbytes = hip.query('crl_embed_bwd_ws_bytes', B, T, dt.shape[1])
    ws = _scratch.get(nbytes, dt.device)
    hip.call('crl_embed_bwd', _p(ids), _p(dt), _p(dtok), _p(dpos), int(accumulate), B, T, dt.shape[1], pos_offset, dtok.shape[0],
             _p(ws), nbytes, _stream())


def cross_entropy(logits, target, V: int, loss_mul: float, grad_mul: float, loss, n_valid, row_loss, dlogits,
                  grad_mul_dev: Optional[torch.Tensor] = None) -> None:
    """grad_mul_dev: device fp32 scalar multiplied into the gradient on the device (the GradScaler loss scale)"""
    M = logits.shape[0]
    hip.call('crl_cross_entropy', _p(logits), logits.stride(0), _p(target), M, V, float(loss_mul), float(grad_mul), _p(grad_mul_dev),
             _p(loss), _p(n_valid), _p(row_loss), _p(dlogits), _stream())


def grad_norm(g, max_norm: float, inv_scale: float, state) -> None:
    ws = _scratch.get(hip.query('crl_grad_norm_ws_bytes'), g.device)
    hip.call('crl_grad_norm', _p(g), g.numel(), float(max_norm), float(inv_scale), _p(state), _p(ws), _stream())


def grad_norm_scaled(g, max_norm: float, grad_divisor: float, growth_factor: float, backoff_factor: float, growth_interval: int,
                     state) -> None:
    """unscale + inf check + clip coefficient + GradScaler.update() on the device (state: 8 floats, scale in state[4])"""
    assert state.numel() >= 8
    ws = _scratch.get(hip.query('crl_grad_norm_ws_bytes'), g.device)
    hip.call('crl_grad_norm_scaled', _p(g), g.numel(), float(max_norm), float(grad_divisor), float(growth_factor), float(backoff_factor),
             int(growth_interval), _p(state), _p(ws), _stream())


def optim_prepare(state, base_lr: float, warmup_lr_init: float, lr_min: float, warmup_t: int, t_initial: int, beta1: float, beta2: float) -> None:
    """device-side learning rate (state[8], cosine + warm-up at state[7] updates; t_initial <= 0: constant) and bias corrections (state[9], [10])"""
    assert state.numel() >= 16
    hip.call('crl_optim_prepare', _p(state), float(base_lr), float(warmup_lr_init), float(lr_min), int(warmup_t), int(t_initial),
             float(beta1), float(beta2), _stream())


def adamw(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step: int, state, zero_grad: bool) -> None:
    """step >= 1: host-side step number; step == 0: device-side mode (bias corrections, and the learning rate when lr < 0, from the
    state words written by optim_prepare)"""
    hip.call('crl_adamw', _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), float(lr), float(beta1), float(beta2),
             float(eps), float(weight_decay), int(step), _p(state), int(zero_grad), _stream())


def cast_bf16(src, dst) -> None:
    hip.call('crl_cast_bf16', _p(src), _p(dst), src.numel(), _stream())


def cast_pad_bf16(src, dst, R: int, C: int, Cp: int) -> None:
    hip.call('crl_cast_pad_bf16', _p(src), _p(dst), R, C, Cp, _stream())


class DropSpec:
    """(p, seed, step) of one forward / backward pair; site ids are chosen by the caller (one per dropout call site)"""

    def __init__(self, p: float, seed: int, step: int, p_attn: float = 0.0, p_act: float = 0.0, p_path: float = 0.0):
        """p: hidden-state dropout; p_attn: attention-probability dropout; p_act: activation (post-GELU) dropout; p_path: the
        encoder's maximum drop-path rate (Swin: per block linearly from 0 to p_path)"""
        self.p, self.seed, self.step = float(p), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 32 - 1)
        self.p_attn, self.p_act, self.p_path = float(p_attn), float(p_act), float(p_path)


def dropout(x: torch.Tensor, y: torch.Tensor, d: DropSpec, site: int, y_bf16: Optional[torch.Tensor] = None, p: Optional[float] = None) -> None:
    """y = dropout(x) (bf16 or fp32, in place allowed); fp32 inputs may emit a bf16 copy of the result. p overrides d.p (activation dropout)"""
    assert x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()
    hip.call('crl_dropout', _p(x), _p(y), x.numel(), int(x.dtype == F32), _p(y_bf16), d.p if p is None else float(p), d.seed, d.step, site, _stream())


def droppath_scale(scale: torch.Tensor, p: float, d: DropSpec, site: int) -> None:
    """scale[b] = keep(b) / (1 - p): one drop-path decision per sample for (d.seed, d.step, site)"""
    _chk(scale, F32, 'droppath scale')
    hip.call('crl_droppath_scale', _p(scale), scale.numel(), float(p), d.seed, d.step, site, _stream())


def rowscale_add(x_bf16: torch.Tensor, scale: torch.Tensor, resid: torch.Tensor, out: torch.Tensor, rows_per_sample: int) -> None:
    """out(f32) = resid + bf16(x * scale[sample of the row])"""
    M, C = x_bf16.shape
    assert x_bf16.is_contiguous() and resid.is_contiguous() and out.is_contiguous()
    hip.call('crl_rowscale_add', _p(x_bf16), _p(scale), _p(resid), _p(out), M, int(rows_per_sample), C, _stream())


def rowscale_bf16(x_bf16: torch.Tensor, scale: torch.Tensor, y_bf16: torch.Tensor, rows_per_sample: int) -> None:
    M, C = x_bf16.shape
    assert x_bf16.is_contiguous() and y_bf16.is_contiguous()
    hip.call('crl_rowscale_bf16', _p(x_bf16), _p(scale), _p(y_bf16), M, int(rows_per_sample), C, _stream())


def dropout_add(x_bf16: torch.Tensor, resid: torch.Tensor, out: torch.Tensor, d: DropSpec, site: int) -> None:
    """out(f32) = resid(f32) + bf16(dropout(x_bf16))"""
    _chk(x_bf16, BF16, 'dropout_add x')
    _chk(resid, F32, 'dropout_add resid')
    assert x_bf16.is_contiguous() and resid.is_contiguous() and out.is_contiguous()
    hip.call('crl_dropout_add', _p(x_bf16), _p(resid), _p(out), x_bf16.numel(), d.p, d.seed, d.step, site, _stream())


def dropout_mask(n: int, d: DropSpec, site: int, device, p: Optional[float] = None) -> torch.Tensor:
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    hip.call('crl_dropout_mask', _p(keep), n, d.p if p is None else float(p), d.seed, d.step, site, _stream())
    return keep


def add_bf16_to_f32(x_bf16, y, accumulate: bool) -> None:
    hip.call('crl_add_bf16_to_f32', _p(x_bf16), _p(y), x_bf16.numel(), int(accumulate), _stream())


# ------------------------------------------------------------------------------------------- measurement aids
def gemm_set_schedule(dynamic: bool) -> None:
    """persistent GEMMs: dynamic tile tickets (default) or the static tile walk (A/B runs, tests); results are bit-identical"""
    hip.call('crl_gemm_set_schedule', int(bool(dynamic)))


def gemm_set_reserved_cus(n: int) -> None:
    """persistent GEMMs launch on 256 - n CUs (the rest is left to RCCL's kernels while gradient buckets are in flight)"""
    hip.call('crl_gemm_set_reserved_cus', int(n))


class OccupyCUs:
    """`with OccupyCUs(n):` -- n CUs are held by a sleeping side-stream kernel (crl_debug_occupy_cus) for the duration of the block:
    the single-GPU stand-in for the CUs RCCL's all-reduce kernels take from the training step of a data-parallel run.
    Inside the block wait for work with `torch.cuda.current_stream().synchronize()`: a device-wide synchronize waits for the sleepers."""

    def __init__(self, n_cus: int, max_seconds: float = 60.0):
        self.n, self.max_seconds = int(n_cus), float(max_seconds)

    def __enter__(self):
        import time
        self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        hip.call('crl_debug_occupy_cus', self.n, self.max_seconds, self.flag.data_ptr(), self.stream.cuda_stream)
        time.sleep(0.02)          # the sleepers are resident before the disturbed work is enqueued
        return self

    def __exit__(self, *exc):
        self.flag[0] = 1
        self.stream.synchronize()
        return False
