/* libcruller_hip.so -- C-ABI of the MI355X-native Cruller pretrain step (gfx950 only).
 *
 * The reference (huggingface/pixparse) has NO FFI/operator layer of its own: every op below is
 * executed for it by timm / transformers / torch.  Each entry point therefore cites
 * the reference call site whose arithmetic it replaces, and the third-party module that executes
 * that arithmetic for the reference.  "ref:" paths are relative to /root/reference/src/pixparse;
 * "hf:" = transformers/models/bart/modeling_bart.py (5.15.0).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (torch tensor .data_ptr()); nothing here allocates,
 *     synchronises or touches the host; `stream` is a hipStream_t passed as void*.
 *   - return 0 on success, <0 on error (crl_last_error() gives the text, thread-local).
 *   - bf16 = raw uint16 storage; "f32"/"bf16" in a name is the storage type of that argument.
 *   - row-major everywhere; ld* are leading dimensions in ELEMENTS.
 */
#ifndef CRL_H
#define CRL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int crl_version(void);
const char* crl_last_error(void);

/* ---------------------------------------------------------------- GEMM (bf16 MFMA, fp32 accumulate)
 * replaces torch.nn.functional.linear under autocast(bf16) and its autograd:
 *   ref: models/image_encoder_timm.py:35-42 (timm Attention.qkv/proj, Mlp.fc1/fc2, PatchEmbed conv,
 *        PatchMerging.reduction), hf:180-183,227-231,252 (q/k/v/out_proj), hf:381-386 (fc1/fc2),
 *        hf:1297 (lm_head).
 * layout:
 *   CRL_NT  C[M,N] = A[M,K] * B[N,K]^T     forward  (B = nn.Linear weight)
 *   CRL_NN  C[M,N] = A[M,K] * B[K,N]       dgrad    (A = dY, B = weight)
 *   CRL_TN  C[M,N] = A[K,M]^T * B[K,N]     wgrad    (A = dY, B = X; contraction over rows)
 * K must be a multiple of 32 for NT/NN (pad the operand); any K for TN. N and ldc multiples of 4,
 * lda/ldb multiples of 8, operands 16-byte aligned and smaller than 4 GiB.
 * epilogue (v = fp32 accumulator, b = bf16-rounded bias, all optional pointers may be NULL):
 *   CRL_EPI_BF16        C(bf16)  = v + b
 *   CRL_EPI_BF16_GELU   h = bf16(v + b);  C(bf16) = gelu_erf(h);  aux(fp16) = gelu_erf'(h)   (saved for the backward: one erf serves both)
 *   CRL_EPI_BF16_DGELU  C(bf16)  = bf16(v) * aux                (aux = the derivative the forward saved, fp16 -- torch's gelu_backward
 *                       evaluates gelu'(h) again from the saved h; here it is rounded once to fp16, 2^-11 relative, a quarter of the
 *                       bf16 rounding of the product)
 *   CRL_EPI_F32_RESID   C(f32)   = resid(f32) + float(bf16(v + b))   (resid may alias C)
 *   CRL_EPI_F32         C(f32)   = v            (beta = 0)
 *   CRL_EPI_F32_ACC     C(f32)  += v            (beta = 1: grad accumulation)
 */
enum { CRL_NT = 0, CRL_NN = 1, CRL_TN = 2 };
enum { CRL_EPI_BF16 = 0, CRL_EPI_BF16_GELU = 1, CRL_EPI_BF16_DGELU = 2, CRL_EPI_F32_RESID = 3,
       CRL_EPI_F32 = 4, CRL_EPI_F32_ACC = 5 };
/* ws (optional, >= crl_gemm_ws_bytes(...)): fp32 scratch that lets the wgrad layout split its long
 * contraction over several workgroups per output tile (deterministic slab reduce, no atomics); the NT / NN
 * layouts use it for the few remainder rows of the wave-quantisation split when K >= 2048 (0 = not needed). */
size_t crl_gemm_ws_bytes(int layout, int epilogue, int64_t M, int64_t N, int64_t K);
/* kernel selection override for tests / A-B runs: 0 = auto (the 256-row kernels when they fill the chip, else 128x128),
 * 1 = always 128x128, 2 = a 256x256 kernel (one workgroup per CU; which one: crl_gemm_set_big_kernel) whenever legal. */
int crl_gemm_set_policy(int policy);
/* which 256x256 kernel serves the big launches: 2 (default) = per launch -- the 4-wave one-wave-per-SIMD kernel with the hand-placed main loop
 * (gemm4w.hip, round 5) where a workgroup walks >= 32 K tiles behind a store-only epilogue (weight gradients, long-K dgrads), else the 8-wave
 * 8-phase kernel (gemm256.hip); 1 / 0 force one of them.  Same results bit for bit; tests / same-box A-B.  Process-wide. */
int crl_gemm_set_big_kernel(int which);
/* 1 (default): plain-bf16 NT / NN launches of the 4-wave kernel with whole column tiles and at least three rounds of tiles run the epilogue of output
 * tile T inside the main loop of tile T + 1 (gemm4w.hip, overlapped form); 0: the classic epilogue between the tiles; 7: the overlapped form for
 * ANY number of tiles (tests).  Same results bit for bit.  Process-wide. */
int crl_gemm_set_overlap(int on);
/* tuning aid for the wave-quantisation cut (gemm.hip quant_rows): multiplies the modelled cost of the remainder launch (default 1);
   < 0 = never cut.  Process-wide. */
int crl_gemm_set_quant_cost(float c);
/* Wave-quantisation cost model: microseconds per round of 256x256 tiles = a + b K / 1024 (defaults fitted on one MI355X).
 * crl_gemm_calibrate: one-off and SYNCHRONISING -- times one / two rounds at K = 1024 / 4096 on random bf16 operands placed in ws
 * (>= crl_gemm_calibrate_ws_bytes(), ~340 MB) and refits a, b for THIS device; 0 = refitted, 1 = implausible measurement (defaults kept). */
size_t crl_gemm_calibrate_ws_bytes(void);
int crl_gemm_calibrate(void* ws, size_t ws_bytes, void* stream);
int crl_gemm_model(float* round_a_us, float* round_b_us, int* calibrated);
/* Data-parallel runs share the GPU with RCCL's all-reduce kernels (ref: DistributedDataParallel's bucket all-reduces,
 * task/task_cruller_pretrain.py:181-189, overlapping backward).  The 256-row kernels are persistent: one (two) resident
 * workgroup(s) per CU.
 * crl_gemm_set_schedule(1) (default): the resident workgroups PULL output tiles from per-launch device ticket counters (one
 *   list per XCD, stealing when a list is empty), so a launch that finds n CUs occupied by another kernel takes
 *   ceil(tiles / (256 - n)) tile times; 0 = the static walk (workgroup b owns tiles b, b + grid, ...: every workgroup that
 *   cannot be placed at once delays the launch by its whole list).  Results are bit-identical either way.
 * crl_gemm_set_reserved_cus(n), 0 <= n <= 224: launch on 256 - n CUs and re-plan the wave-quantisation split for that
 *   width (set by the gradient reducer while buckets are in flight when PIXPARSE_AMD_RCCL_CUS asks for it; default 0). */
int crl_gemm_set_schedule(int dynamic);
int crl_gemm_set_reserved_cus(int n);
int crl_gemm_bf16(int layout, int epilogue, int64_t M, int64_t N, int64_t K,
                  const void* A, int64_t lda, const void* B, int64_t ldb,
                  const float* bias, void* C, int64_t ldc, void* aux, int64_t ldaux,
                  const float* resid, int64_t ldr, float colscale, int64_t colscale_cols, void* ws, size_t ws_bytes, void* stream);
/* CRL_TN (weight gradient C[M, N] (+)= A[K, M]^T B[K, N]) with aux != NULL (round 6): aux is float[M] and receives the COLUMN SUMS of A,
 * aux[m] (+)= sum_k A[k][m] (ldaux != 0: accumulate) -- the bias gradient of the Linear whose weight gradient this is (ref: autograd of
 * F.linear: grad_bias = grad_output.sum(0)) from the kernel that streams dY anyway: the 4-wave kernel multiplies its A fragments with a fragment
 * of ones on the matrix pipe (exact fp32 sums of the bf16 values, deterministic order); when another kernel serves the launch the stand-alone
 * column-sum pass (crl_colsum_bf16) runs inside the call.  Needs ws >= crl_gemm_ws_bytes(CRL_TN, ...).  fp32 epilogues only. */
/* colscale / colscale_cols (CRL_EPI_BF16 only; 0 columns = off): C[:, 0:colscale_cols] = bf16((v + b) * colscale) -- ONE rounding.  Lets
 * the q part of a q|k|v projection leave the GEMM as q * softmax_scale * log2(e), which is what crl_attn_fwd / crl_attn_bwd take
 * with q_prescaled = 1 (timm Attention: q * self.scale; SDPA: the scale argument): the flash kernels then get base-2 logits straight
 * out of the Q.K^T MFMAs, with no multiply per score. */

/* Single-query attention over a KV cache (generation; replaces F.scaled_dot_product_attention inside transformers'
 * BartAttention for a [B, 1, D] query, reference utils/ocr_utils.py:181-187 -> text_decoder_hf.py:39-45):
 * q [B, H*64] (row stride q_bs), k / v [B, Nk, H*64] strided views (batch stride, row stride in elements), o [B, H*64]
 * bf16. Keys are split over workgroups (HBM-bound); ws >= crl_attn_decode_ws_bytes(B, H, Nk) bytes of scratch.
 * nk_minus1_dev (optional, device int): the valid prefix is *nk_minus1_dev + 1 keys and Nk is the cache capacity -- lets
 * one captured hipGraph serve every step of the generation loop. q_row_dev (optional): q is advanced by *q_row_dev *
 * q_row_stride elements (the query of the step sits in its cache row). */
size_t crl_attn_decode_ws_bytes(int B, int H, int Nk);
int crl_attn_decode(const void* q, int64_t q_bs, const void* k, int64_t k_bs, int64_t k_rs, const void* v, int64_t v_bs,
                    int64_t v_rs, void* o, int64_t o_bs, int B, int H, int Nk, float scale, const int* nk_minus1_dev,
                    const int* q_row_dev, int64_t q_row_stride, void* ws, size_t ws_bytes, void* stream);

/* Skinny linear layer for generation (replaces nn.Linear / F.linear on [B, 1, K] decode-step activations inside
 * transformers' BartDecoderLayer, reached from the reference through models/text_decoder_hf.py:39-45 and
 * utils/ocr_utils.py:181-187):  out[M, N] = epilogue(x[M, K] @ W[N, K]^T + bias),  1 <= M <= 16, N % 4 == 0, K % 8 == 0.
 * epilogue: CRL_EPI_BF16 (bf16 out), CRL_EPI_BF16_GELU (bf16 GELU(out)), CRL_EPI_F32_RESID (fp32 out = resid + bf16-rounded
 * result). Same arithmetic as crl_gemm_bf16 (fp32 accumulate, bias rounded to bf16); HBM-bound on W.
 * out_row_dev (optional, device int): out is advanced by *out_row_dev * out_row_stride elements (the KV-cache row of the
 * current step, so that the launch is identical for every step of a captured generation graph). */
int crl_linear_skinny_bf16(int epilogue, int M, int64_t N, int64_t K, const void* x, int64_t ldx, const void* W,
                           int64_t ldw, const float* bias, void* out, int64_t ldo, const float* resid, int64_t ldr,
                           const int* out_row_dev, int64_t out_row_stride, void* stream);
/* The same with the LayerNorm in FRONT of the projection fused in (post-LN BART: hidden = LayerNorm(residual sum); the next
 * projection reads bf16(hidden), the next residual add reads hidden):  h = LayerNorm(x_f32[M, K]; gamma, beta, eps) with the arithmetic of
 * crl_layernorm_fwd (bit-identical), out = epilogue(bf16(h) @ W^T + bias); h_f32 (optional, needs N >= K) receives the fp32 h.
 * Every workgroup normalises the <= 16 rows for itself (L2 reads) -- one launch per LayerNorm less on the decode step's dependent chain.
 * 64 <= K <= 2048; epilogue CRL_EPI_BF16 or CRL_EPI_BF16_GELU. */
int crl_linear_skinny_ln_bf16(int epilogue, int M, int64_t N, int64_t K, const float* x_f32, int64_t ldx, const float* gamma,
                              const float* beta, float eps, float* h_f32, int64_t ldh, const void* W, int64_t ldw,
                              const float* bias, void* out, int64_t ldo, const int* out_row_dev, int64_t out_row_stride,
                              void* stream);

/* column sums of a bf16 matrix into fp32 (bias gradients): out[n] (+)= sum_m X[m,n].
 * ws: >= crl_colsum_ws_bytes(N) bytes of scratch. */
size_t crl_colsum_ws_bytes(int64_t N);
int crl_colsum_bf16(const void* X, int64_t M, int64_t N, int64_t ldx, float* out, int accumulate,
                    void* ws, void* stream);

/* ---------------------------------------------------------------- LayerNorm (fp32 math, wavefront reduce)
 * replaces F.layer_norm (autocast keeps it fp32): timm Block.norm1/norm2/norm_pre/norm,
 * hf:652 (layernorm_embedding), hf:363,378,388 (post-LN).  y_f32 and/or y_bf16 may be NULL. */
int crl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps,
                      int64_t M, int64_t D, float* y_f32, void* y_bf16, float* mean, float* rstd,
                      void* stream);
/* dy = dy_f32 (optional) + float(dy_bf16) (optional).  dx_f32 (+)= LN'(dy) when dx_accumulate;
 * dx_bf16 (optional) receives a bf16 copy of the final value written to dx_f32 (the accumulated
 * sum when dx_accumulate) -- it is the gradient the next GEMM backward consumes.
 * dx_colsum (optional, [D]): column sums over the M rows of the bf16 values written to dx_bf16 = the bias gradient of the
 *   Linear whose output gradient dx_bf16 is (nn.Linear backward: db = sum_rows dY) -- fused here because the rows pass
 *   through registers anyway; replaces a crl_colsum_bf16 pass over dx_bf16.
 * dgamma/dbeta/dx_colsum are accumulated (+=) when acc_wgrad else overwritten.  ws >= crl_layernorm_bwd_ws_bytes(D). */
size_t crl_layernorm_bwd_ws_bytes(int64_t D);
int crl_layernorm_bwd(const float* dy_f32, const void* dy_bf16, const float* x, const float* gamma,
                      const float* mean, const float* rstd, int64_t M, int64_t D,
                      float* dx_f32, int dx_accumulate, void* dx_bf16,
                      float* dgamma, float* dbeta, float* dx_colsum, int acc_wgrad, void* ws, void* stream);

/* ---------------------------------------------------------------- flash attention, head_dim 64
 * replaces F.scaled_dot_product_attention: timm Attention (ViT global MHSA, non-causal),
 * hf:185-257 (decoder self-attention is_causal=True, cross attention unmasked).
 * element (b, n, h, j) of q lives at q + b*q_bs + n*q_rs + h*64 + j (strides in elements), same
 * for k, v, o and their gradients; lse is [B, H, Nq] fp32 (natural log).  causal aligns the
 * diagonal bottom-right (key j visible to query i iff j <= i + Nk - Nq). */
int crl_attn_fwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                 const void* v, int64_t v_bs, int64_t v_rs, void* o, int64_t o_bs, int64_t o_rs,
                 float* lse, int B, int H, int Nq, int Nk, float scale, int causal, int q_prescaled,
                 float drop_p, uint64_t drop_seed, uint32_t drop_step, uint32_t drop_site, void* stream);
/* q_prescaled = 1: q holds q * scale * log2(e) (crl_gemm_bf16 colscale on the q columns of the projection).  The forward then runs the
 * seeded / lazy-maximum kernel (base-2 logits straight from the Q.K^T MFMAs; `scale` unused); the backward needs the same flag AND the
 * true `scale` (it stores dQ as the gradient of the unscaled projection output, as without prescaling).  lse is the natural-log
 * log-sum-exp of the true scores either way -- with ONE documented deviation: the hand-placed forward stream (non-causal, prescaled q, Nk >= 128:
 * every training attention of the ViT encoders and the cross-attentions) takes l as the sum of the bf16-ROUNDED probabilities, the operand of
 * P.V (row sums on the matrix pipe), so that the output's weights sum to one exactly; its lse therefore differs from the fp32 log-sum-exp by at
 * most log(1 + 2^-8) = 3.9e-3 (rows that one key dominates; random rows: ~1e-4), and the probabilities the backward rebuilds as exp2(s - lse)
 * sum to 1 within that factor instead of within fp32 rounding.  Bounded by tests (test_attention_fwd_one_wave_per_simd: 5e-3;
 * test_attention_stream_lse_on_peaked_rows); crl_attn_fwd_set_mode(1) selects the kernels with the fp32 row sum. */
/* delta: scratch of 2*B*H*Nq fp32 (row constants -delta = -rowsum(dO o O) and -lse/scale that seed the MFMA accumulators).
 * dq/dk/dv strides as q/k/v.  Two forms (same results up to bf16 rounding of dq; both deterministic, no float atomics):
 *   two-pass    dQ pass (recomputes S, dP; produces the row constants) then dK/dV pass (recomputes S, dP): 7 MFMA products per tile;
 *   single pass (non-causal, ws >= crl_attn_bwd_ws_bytes(...) > 0): one recomputation feeds dK, dV and dQ -- 5 products; a
 *               workgroup = 4 waves owns 256 keys of a head (one wave per SIMD with the whole register file: dK^T / dV^T accumulators,
 *               K / V row fragments and the K^T fragments of its dQ block stay in registers; only dS crosses LDS), so dQ is a sum over
 *               the key blocks: every block writes its partial as a bf16 [B, Nq, H*64] slab into ws and a reduce pass adds the
 *               ceil(Nk / 256) slabs in fixed order in fp32, applies `scale` and rounds to dq.  With q_prescaled the kernel is one
 *               hand-placed instruction stream (csrc/gen_attn_bwd_sp.py); without, the C++ form of the same algorithm (slow: tests).
 * crl_attn_bwd_ws_bytes: 0 when the two-pass form will run.
 * crl_attn_bwd_set_mode: 0 = auto: single pass for non-causal problems with Nq >= 1000, Nk >= 1024 and a prescaled q (the ViT encoders:
 *   same-box 3.22 against 3.51 ms per layer at B 8, H 16, N 6189; cross-attention 1023 x 6189: 0.63 against 0.67), two-pass otherwise; 1 = two-pass; 2 = single pass whenever legal
 *   (non-causal); 3 = single pass in its C++ form (reference of the hand-placed stream: bit-identical results). */
/* crl_attn_fwd with q_prescaled, no causal mask, no dropout and Nk >= 128 runs a hand-placed instruction stream (csrc/gen_attn_fwd4w.py:
 * 256 queries per workgroup, 64 per wave, so that every K / V fragment read from LDS feeds two MFMAs; software pipeline over quarter tiles; row
 * sums of the bf16 probabilities on the matrix pipe; the softmax reference of a row is its exact maximum over the first key tile; a block with
 * a non-finite row is re-run with the moving-maximum kernel).  Two register budgets of the same pipeline: 256 registers per wave = two
 * workgroups per CU (default: two waves per SIMD overlap each other's v_exp / LDS / MFMA issue and the stream runs at the matrix pipe's rate)
 * and 512 = one per CU.  crl_attn_fwd_set_mode: 0 = auto (default), 1 = the 32-queries-per-wave kernels everywhere, 2 = the stream with its
 * fallback forced on every block (tests), 3 = the 512-register form, 4 = the 256-register form. */
int crl_attn_fwd_set_mode(int mode);
/* 1 (default): with more query blocks than workgroup slots the stream is launched persistently and pulls its blocks from the per-XCD ticket lists
 * (as crl_attn_bwd_set_persistent); 0 = one workgroup per block.  Same results. */
int crl_attn_fwd_set_persistent(int on);
size_t crl_attn_bwd_ws_bytes(int B, int H, int Nq, int Nk, int causal);
int crl_attn_bwd_set_mode(int mode);
/* Single pass only: key blocks per workgroup.  A workgroup walks `chain` consecutive 256-key blocks of its (batch, head) and adds each
 * block's partial dQ to what the blocks before it left in the slab (read back tile by tile as the C operand of the tile's first dQ MFMA),
 * so the reduce adds ceil(ceil(Nk / 256) / chain) slabs instead of ceil(Nk / 256).  0 (default) = chosen per problem from the simulated
 * makespan of the workgroups on the CUs not reserved for RCCL (crl_gemm_set_reserved_cus); n >= 1 forces n (tests, A/B).  dK / dV do not
 * depend on it; dQ carries one more bf16 rounding of the running sum per link (3.5e-3 instead of 2.6e-3 relative L2 against fp32 at
 * chain 4, 25 key blocks).  crl_attn_bwd_ws_bytes stays sized for chain 1. */
int crl_attn_bwd_set_chain(int chain);
/* Single pass only: the key blocks a head has left over after its full chains (nkt % chain) are walked by TWO workgroups that take half of the query
 * tiles each when that shortens the launch (cfg-3: 3200 key blocks over 256 CUs = 12.5 per CU -- without the split half the CUs walk 13): dQ rows are
 * disjoint, the second half's dK / dV go to a scratch behind the slabs and are added to the first half's by a small kernel (one more bf16 rounding on
 * those key rows).  -1 (default) = decided with the automatic chain from the simulated makespan; 0 = never; 1 = whenever legal (tests). */
int crl_attn_bwd_set_qsplit(int mode);
/* 1 when the hand-placed single pass would split the remainder chains for Nk keys and BH = B * H heads under the current settings; host arithmetic only */
int crl_attn_bwd_qsplit_for(int Nk, int BH);
/* Single pass only: 1 (default) = when there are more chains than CUs the launch is persistent -- one workgroup per CU (minus the CUs reserved by
 * crl_gemm_set_reserved_cus) pulls chains from the per-XCD ticket lists of the persistent GEMMs (crl_gemm_set_schedule(0) switches both to the
 * static walk) and steals from the other XCDs' lists at the end; 0 = one workgroup per chain.  Same results either way. */
int crl_attn_bwd_set_persistent(int on);
/* the chain length the hand-placed single pass would use for Nk keys and BH = B * H heads under the current settings (forced chain,
 * reserved CUs); host arithmetic only */
int crl_attn_bwd_chain_for(int Nk, int BH);
int crl_attn_bwd(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs,
                 const void* v, int64_t v_bs, int64_t v_rs, const void* o, int64_t o_bs, int64_t o_rs,
                 const void* d_o, int64_t do_bs, int64_t do_rs, const float* lse, float* delta,
                 void* dq, int64_t dq_bs, int64_t dq_rs, void* dk, int64_t dk_bs, int64_t dk_rs,
                 void* dv, int64_t dv_bs, int64_t dv_rs,
                 int B, int H, int Nq, int Nk, float scale, int causal, int q_prescaled,
                 float drop_p, uint64_t drop_seed, uint32_t drop_step, uint32_t drop_site, void* ws, size_t ws_bytes, void* stream);
/* drop_p > 0: attention-probability dropout (transformers BartAttention / SDPA dropout_p = config.attention_dropout, hf:240-252:
 * softmax, THEN dropout of the probabilities, then P.V).  Element (b, h, q, k) is kept iff a 32-bit hash of its index keyed by
 * (drop_seed, drop_step, drop_site) clears the 24-bit threshold p; the mask is never stored -- the backward passes re-evaluate it (the
 * two-pass form only).  crl_attn_dropout_mask writes the same mask as bytes [B, H, Nq, Nk] (tests: the oracle applies the kernels' mask). */
int crl_attn_dropout_mask(void* keep_u8, int B, int H, int Nq, int Nk, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream);

/* Live per-kernel timing for bench.py's roofline object: between crl_prof_begin and crl_prof_end every launch of an
 * instrumented kernel is bracketed by HIP events on its own stream (pool of `capacity` pairs created up front, no
 * synchronisation until the end). crl_prof_end fills, per kernel id < n_ids, the number of launches, their summed
 * duration (ms) and their summed algorithmic work (FLOPs). Kernel ids: */
#define CRL_K_ATTN_FWD 0        /* attn_fwd_kernel<false>,      + 1 = <true> (causal) */
#define CRL_K_ATTN_BWD_DKDV 2   /* attn_bwd_dkdv_kernel<false>, + 1 = <true> */
#define CRL_K_ATTN_BWD_DQ 4     /* attn_bwd_dq_kernel<false>,   + 1 = <true> */
#define CRL_K_ATTN_BWD_FUSED 6  /* attn_bwd_spx_kernel / attn_bwd_sp_kernel (single-pass backward; work = the whole algorithmic backward) */
#define CRL_K_ATTN_DQ_REDUCE 7  /* attn_dq_reduce_kernel (sum of the partial-dQ slabs; no FLOPs credited) */
#define CRL_K_COUNT 8
/* Measurement aid (never on the product path): n_cus workgroups that each take a whole CU (all 160 KiB of its LDS) and sleep
 * until max_seconds (<= 120) have passed or *stop_flag (optional; device-visible, e.g. pinned host memory) becomes non-zero.
 * Stands in, on one GPU, for the CUs RCCL's all-reduce kernels hold while gradient buckets are in flight (bench.py
 * --occupy-cus). Launch it on a stream of its own BEFORE the work it should disturb. */
int crl_debug_occupy_cus(int n_cus, double max_seconds, const int* stop_flag, void* stream);
int crl_prof_begin(int capacity);
int crl_prof_end(int n_ids, int* launches, double* ms, double* work);

/* measurement hook (bench.py times the three backward launches one by one): bit0 = delta, bit1 = dK/dV pass,
 * bit2 = dQ pass; default 7 = all. */
int crl_attn_bwd_set_parts(int parts);

/* ---------------------------------------------------------------- Swin shifted-window attention
 * replaces timm WindowAttention + roll / window_partition / window_reverse
 * (cross-check: transformers/models/swin/modeling_swin.py:343-370,486-505,529-626).
 * qkv: [B, Hf, Wf, 3, heads, hd] bf16 in natural NHWC token order (output of the qkv Linear),
 * out: [B, Hf, Wf, heads*hd] bf16 in natural order; the cyclic shift, the window partition and
 * the 0/-100 region mask are index arithmetic inside the kernel.  table: [(2w-1)^2, heads] fp32.
 * hd must be 32 (every timm Swin). w*w <= 64. */
int crl_swin_attn_fwd(const void* qkv, const float* table, void* out, int B, int Hf, int Wf,
                      int heads, int w, int shift, float scale, void* stream);
/* dtable accumulates (+=) with fp32 atomics when acc else must be zeroed by the caller first. */
int crl_swin_attn_bwd(const void* qkv, const float* table, const void* d_out, void* dqkv,
                      float* dtable, int B, int Hf, int Wf, int heads, int w, int shift, float scale,
                      void* stream);
/* PatchMerging gather: x [B,Hf,Wf,C] f32 -> y [B,Hf/2,Wf/2,4C] f32, order [(0,0),(1,0),(0,1),(1,1)];
 * bwd scatters dy back (pure permutation). */
int crl_patch_merge_fwd(const float* x, float* y, int B, int Hf, int Wf, int C, void* stream);
int crl_patch_merge_bwd(const float* dy, float* dx, int B, int Hf, int Wf, int C, void* stream);

/* ---------------------------------------------------------------- patch embedding glue
 * im2row for the stride==kernel patch conv (timm PatchEmbed): image [B,C,H,W] f32 ->
 * patches [B*gh*gw, Kp] bf16, column order (c, ph, pw) = conv weight.flatten(1), columns
 * C*P*P..Kp-1 zero.  Only the top-left gh*P x gw*P pixels are read (floor crop). */
int crl_im2row(const float* image, void* patches, int B, int C, int H, int W, int P, int gh, int gw,
               int Kp, void* stream);
/* ViT token assembly: x[b,0,:] = cls + pos[0]; x[b,1+i,:] = float(patch[b,i,:]) + pos[1+i]  (f32). */
int crl_vit_tokens_fwd(const void* patch_bf16, const float* cls, const float* pos, float* x,
                       int B, int Np, int D, void* stream);
/* bwd: dpatch(bf16) = dx[b,1+i]; dpos (+)= sum_b dx[b]; dcls (+)= sum_b dx[b,0]. */
int crl_vit_tokens_bwd(const float* dx, void* dpatch_bf16, float* dcls, float* dpos, int acc,
                       int B, int Np, int D, void* stream);

/* ---------------------------------------------------------------- decoder embedding
 * hf:609,648-651: t = embed_tokens[ids] * 1.0 + embed_positions[arange(T) + 2]  -> f32 [B*T, D].
 * (layernorm_embedding is then crl_layernorm_fwd).  ids int64. */
int crl_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* out,
                  int B, int T, int D, int pos_offset, int vocab, void* stream);
/* generation: one token per sequence at position *step_dev (device int) -> f32 [B, D] */
int crl_embed_decode(const int64_t* ids, const float* tok, const float* pos, float* out, int B, int D, int pos_offset,
                     int vocab, const int* step_dev, void* stream);
/* dtok[ids] += dt (dtok is the tied LM-head grad), dpos[t+off] (+)= sum_b dt.  Deterministic: the rows of a token that
 * occurs several times are added in the order of their positions (counting rank + segmented sums in ws, no float
 * atomics), so the result does not depend on scheduling.  ws >= crl_embed_bwd_ws_bytes(B, T, D), 16-byte aligned.
 * ids outside [0, vocab) (torch: device assert): forward rows become NaN, backward skips them -- never out of bounds. */
size_t crl_embed_bwd_ws_bytes(int B, int T, int D);
int crl_embed_bwd(const int64_t* ids, const float* dt, float* dtok, float* dpos, int acc_pos,
                  int B, int T, int D, int pos_offset, int vocab, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- shifted-token cross-entropy
 * ref: task/task_cruller_pretrain.py:118,251-256  nn.CrossEntropyLoss(ignore_index=-100) on bf16
 * logits upcast to fp32; mean over targets != -100; optional / accum_steps and GradScaler scale.
 * logits [M, ldl] bf16 (only the first V columns are read).  On return
 *   loss[0] = mean NLL * loss_mul  (fp32), n_valid[0] = number of non-ignored rows (int32),
 *   dlogits (may alias logits) = bf16((softmax - onehot) * grad_mul / n_valid), columns V..ldl-1 zeroed.
 * grad_mul_dev (optional, device): one more factor on the gradient read on the device -- the GradScaler loss scale
 *   that crl_grad_norm_scaled keeps in state[4].  A target outside [0, V) other than -100 makes the loss NaN.
 * row_loss: scratch [M] fp32. */
int crl_cross_entropy(const void* logits, int64_t ldl, const int64_t* target, int64_t M, int V,
                      float loss_mul, float grad_mul, const float* grad_mul_dev, float* loss, int32_t* n_valid,
                      float* row_loss, void* dlogits, void* stream);

/* ---------------------------------------------------------------- optimiser over the flat arenas
 * ref: task/task_cruller_pretrain.py:191-206,259-295 -> timm NativeScaler (GradScaler unscale,
 * inf check), dispatch_clip_grad('norm') = torch clip_grad_norm_, torch.optim.AdamW(wd=0),
 * optimizer.zero_grad().
 * crl_grad_norm: state[0] = ||g||_2 * inv_scale, state[1] = clip coefficient
 *   min(1, max_norm / (norm + 1e-6)) * inv_scale (or inv_scale when max_norm <= 0),
 *   state[2] = 1.0 if any grad is inf/nan else 0.0, state[3] += 1 when the step will be taken (torch's
 *   state['step'] does not advance on a step GradScaler skips).   ws >= crl_grad_norm_ws_bytes(); state: >= 4 floats.
 * crl_grad_norm_scaled: the same with torch.amp.GradScaler kept on the device (state: 8 floats): the loss scale is
 *   state[4], inv_scale = 1 / (state[4] * grad_divisor) (grad_divisor = data-parallel world size: the all-reduce sums),
 *   and after the inf check  state[4] *= backoff_factor, state[5] = 0  on inf/nan, else  state[5] += 1 and every
 *   growth_interval clean steps  state[4] *= growth_factor  (torch _amp_update_scale_): no host synchronisation,
 *   the next crl_cross_entropy reads the updated scale through grad_mul_dev = &state[4]. */
size_t crl_grad_norm_ws_bytes(void);
int crl_grad_norm(const float* g, int64_t n, float max_norm, float inv_scale, float* state, void* ws,
                  void* stream);
int crl_grad_norm_scaled(const float* g, int64_t n, float max_norm, float grad_divisor, float growth_factor,
                         float backoff_factor, int growth_interval, float* state, void* ws, void* stream);
/* p,m,v updated in place with g*state[1]; skipped entirely when state[2] != 0 (GradScaler.step);
 * p_bf16 (optional) receives the bf16 shadow of the new p; g is zeroed when zero_grad != 0.
 * step >= 1: bias corrections 1 - beta^step from the host's count; step == 0 (device-side mode, state: 16 floats): bias
 * corrections from state[9], state[10] and -- when lr < 0 -- the learning rate from state[8], all three written by
 * crl_optim_prepare; in that mode a positive state[6] clamps every unscaled gradient element to [-state[6], state[6]]
 * first (torch clip_grad_value_, timm dispatch_clip_grad mode 'value').
 * crl_optim_prepare (one thread, between crl_grad_norm* and crl_adamw): what the host would otherwise compute per step and
 * pass by value, so that every launch of a train step has identical arguments step after step and the whole step can be
 * replayed from a hipGraph -- state[8] = learning rate of this update = timm CosineLRScheduler(t_initial, lr_min,
 * warmup_t, warmup_lr_init) at u = state[7] updates attempted so far (ref task_cruller_pretrain.py:214-224,292-295;
 * t_initial <= 0: constant base_lr), state[9] = 1 - beta1^t and state[10] = 1 / sqrt(1 - beta2^t) in double precision with
 * t = steps actually taken (this one included), then the update count advances.  The two counts are kept exactly as uint32 in words 11
 * (steps taken, advanced by crl_grad_norm* when the step is not skipped) and 12 (updates attempted) of the state vector; state[3] and
 * state[7] are their fp32 mirrors (exact below 2^24) for host-side readers. */
int crl_optim_prepare(float* state, float base_lr, float warmup_lr_init, float lr_min, int warmup_t, int t_initial,
                      float beta1, float beta2, void* stream);
int crl_adamw(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
              float beta2, float eps, float weight_decay, int step, const float* state, int zero_grad,
              void* stream);
/* fp32 -> bf16 cast of a flat range (initial shadow build), and strided fp32 [R,C] -> bf16 [R,Cp]
 * with zero padded columns (padded K operands). */
int crl_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int crl_cast_pad_bf16(const float* src, void* dst, int64_t R, int64_t C, int64_t Cp, void* stream);
/* y_f32 (+)= float(x_bf16), elementwise (gradient joins). */
int crl_add_bf16_to_f32(const void* x_bf16, float* y, int64_t n, int accumulate, void* stream);

/* ---------------------------------------------------------------- dropout (opt-in, off in parity runs and bench.py)
 * ref: transformers BartDecoder / BartDecoderLayer nn.functional.dropout(hidden_states, p=self.dropout, training=self.training)
 * (modeling_bart.py:362,377,384-386,654): live in the reference only for a decoder built with pretrained=False.
 * Stateless mask: element i keeps its value (scaled by 1/(1-p)) iff the 16-bit Philox4x32-10 uniform of (seed, step, site, i) is
 * >= p * 65536, so the backward pass regenerates it from the same triple.  n: multiple of 8; in place allowed.
 *   crl_dropout      y = dropout(x); bf16 (is_f32 = 0) or fp32 (is_f32 = 1, optional bf16 copy of the result in y_bf16)
 *   crl_dropout_add  out(f32) = resid(f32) + bf16(dropout(x_bf16))        (residual join behind a dropped branch)
 *   crl_dropout_mask keep[i] = 0 / 1 as bytes (tests: lets the CPU oracle apply the identical mask) */
int crl_dropout(const void* x, void* y, int64_t n, int is_f32, void* y_bf16, float p, uint64_t seed, uint32_t step, uint32_t site,
                void* stream);
int crl_dropout_add(const void* x_bf16, const float* resid, float* out, int64_t n, float p, uint64_t seed, uint32_t step,
                    uint32_t site, void* stream);
int crl_dropout_mask(void* keep_u8, int64_t n, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream);
/* Drop-path / stochastic depth (timm DropPath on both residual branches of a Swin block; swin_tiny's default drop_path_rate 0.1 is live
 * in the reference because create_model leaves the encoder in train mode): ONE keep decision per sample and branch.
 * crl_droppath_scale: scale[b] = keep(b) / (1 - p), keep from Philox(counter b; site, step; seed).
 * crl_rowscale_add:   out(f32) = resid + float(bf16(x * scale[row / rows_per_sample]))    -- the residual join behind a dropped branch.
 * crl_rowscale_bf16:  y = bf16(x * scale[row / rows_per_sample])                          -- the gradient entering that branch. */
int crl_droppath_scale(float* scale, int B, float p, uint64_t seed, uint32_t step, uint32_t site, void* stream);
int crl_rowscale_add(const void* x_bf16, const float* scale, const float* resid, float* out, int64_t rows, int64_t rows_per_sample, int64_t C,
                     void* stream);
int crl_rowscale_bf16(const void* x_bf16, const float* scale, void* y_bf16, int64_t rows, int64_t rows_per_sample, int64_t C, void* stream);

/* ---------------------------------------------------------------- image preprocessing
 * ref: task/task_cruller_pretrain.py:132-143  ToTensor -> Resize(image_size, BICUBIC, antialias=True) -> Normalize.
 * img: uint8 [H, W, C] (decoded page); out: fp32 [C, Ho, Wo]; tmp: fp32 scratch [C, H, Wo].
 * (xmin, xsize, xw[Wo][xk]) / (ymin, ysize, yw[Ho][yk]): aten upsample_bicubic2d_aa filter tables for W->Wo / H->Ho. */
int crl_image_preprocess_u8(const void* img_hwc_u8, int H, int W, int C, const int32_t* xmin, const int32_t* xsize,
                            const float* xw, int xk, const int32_t* ymin, const int32_t* ysize, const float* yw, int yk,
                            const float* mean, const float* stdv, float* tmp, float* out_chw, int Ho, int Wo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRL_H */
