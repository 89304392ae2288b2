// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: 128x128 tiles, gemm256.hip: 256x256 8-phase):
// argument block, LDS images + swizzles, LDS-DMA staging and fragment reads.
#pragma once
#include "common.h"

namespace gemmc {


struct GemmArgs {
  const u16* A; const u16* B; const float* bias; void* C; void* aux; const float* resid;
  int M, N, K;
  int lda, ldb, ldc, ldaux, ldr;
  uint32_t a_bytes, b_bytes;
  int ntm, ntn;
  int kchunk;            // K tiles per split (gridDim.y splits; split s writes slab s of C)
  size_t slab_stride;    // elements between slabs
  uint32_t* sched;       // dynamic tile scheduler state of this launch (persistent launches only, else nullptr): see TileSched
  float colscale;        // bf16 epilogue: output columns [0, colscale_cols) are multiplied by colscale before the rounding (0 columns = off);
  int colscale_cols;     // the q part of a fused q|k|v projection leaves the GEMM as q * scale * log2(e) (crl_attn_* with q_prescaled)
  // weight-gradient layout, 4-wave kernel only (gemm4w.hip, gen_gemm4w.py colsum_block): the column sums of the A operand (= the bias gradient) as
  // partial rows cs_ws[((split * cs_ntn + tn) * 2 + wc) * M + m]; the first cs_ntn (1, 2 or 4) column tiles of a row of tiles share the work
  float* cs_ws;
  int cs_ntn;
};

// ---- dynamic tile scheduler of the persistent kernels -------------------------------------------------------------------------
// A persistent launch puts (at most) one workgroup on every CU it can get and lets the RESIDENT workgroups pull output tiles from
// device-side ticket counters, instead of giving workgroup b the fixed list b, b + grid, b + 2 grid, ...: when some CUs are held by
// another kernel (RCCL's all-reduce kernels while gradient buckets are in flight, include/crl.h crl_gemm_set_reserved_cus), the
// workgroups that could not be placed simply find the queue empty when they finally start, and the launch takes
// ceil(tiles / resident CUs) tile times instead of up to twice the undisturbed time.
// State = 16 words: heads[0..7] = next position of XCD x's list, word 8 = workgroups that have left.  XCD x's list is the chunk of
// the XCD-aware tile order that xcd_remap gives to blocks b = x (mod 8) (tiles that share operand panels stay in one L2); a
// workgroup pulls from the list of the XCD it really runs on (HW_REG_XCC_ID) and steals from the next lists once its own is empty.
// Every workgroup leaves through sched_leave(): the last one zeroes the state for the next launch that uses the slot (launches that
// share a slot are ordered by their stream; the host rotates through CRL_SCHED_SLOTS slots).
constexpr int CRL_SCHED_WORDS = 16, CRL_SCHED_SLOTS = 64;
__device__ __forceinline__ int sched_xcd() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u); }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ int sched_list_len(int x, int ntiles) { return ntiles > x ? (ntiles - x + 7) >> 3 : 0; }
__device__ __forceinline__ uint32_t sched_pull(uint32_t* head) {
  return __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// logical position (argument of xcd_remap) of the ticket `i` pulled from list x, or the next list with work left; -1 = all lists are
// empty.  Called by ONE lane; x is advanced to the list that answered so that later pulls start there.
__device__ __forceinline__ int sched_resolve(uint32_t* st, int& x, uint32_t i, int ntiles) {
  for (int tries = 0;;) {
    if ((int)i < sched_list_len(x, ntiles)) return x + 8 * (int)i;
    if (++tries == 8) return -1;
    x = (x + 1) & 7;
    i = sched_pull(st + x);
  }
}
__device__ __forceinline__ void sched_leave(uint32_t* st, uint32_t nwg) {   // one lane per workgroup, after its last pull
  const uint32_t gone = __hip_atomic_fetch_add(st + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (gone == nwg - 1) {
#pragma unroll
    for (int j = 0; j < 9; ++j) __hip_atomic_store(st + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- position in the (XCD-aware) tile order -> tile coordinates.  Round 5: GROUPED order -- positions fill blocks of TILE_GROUP_M tile rows
// column by column, so the ~32 consecutive positions the resident workgroups of one XCD hold at any time form an 8 x 4 block of tiles that
// shares 8 A row panels and 4 B column panels (12 panel fetches per K step into that XCD's L2) instead of a 1 x 32 strip of a row-major order
// (33 panel fetches when the output is 32 tiles wide: at 8192^3 the operand traffic behind the L2s, not the MFMA loop, set the pace).
#ifndef TILE_GROUP_M
#define TILE_GROUP_M 8
#endif
__device__ __forceinline__ void tile_coords(int t, int ntm, int ntn, int& tm, int& tn) {
  if (TILE_GROUP_M <= 1) { tm = t / ntn; tn = t % ntn; return; }
  const int per_group = TILE_GROUP_M * ntn;
  const int grp = t / per_group, r = t - grp * per_group;
  const int first = grp * TILE_GROUP_M;
  const int gm = min(ntm - first, TILE_GROUP_M);
  tm = first + r % gm;
  tn = r / gm;
}

// ---- swizzles (see the bank analysis in DESIGN.md "GEMM LDS images") ----
template <int BK> __device__ __forceinline__ int km_swz(int row) {
  return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}
__device__ __forceinline__ int tr_swz(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

// Issue the LDS-DMA for one operand tile.  KM: tile rows = matrix rows r0.., k contiguous from k0.
template <int BK, int NTHR>
__device__ __forceinline__ void stage_km(__amdgpu_buffer_rsrc_t rs, char* lds, int r0, int k0, int ld,
                                         int tid, int wave) {
  constexpr int CPR = BK / 8;                   // 16-B chunks per row
  constexpr int NCH = 128 * CPR / NTHR;   // chunks per thread
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = it * NTHR + tid;
    const int row = c / CPR, ps = c % CPR;
    const int ls = ps ^ km_swz<BK>(row);
    const uint32_t off = ((uint32_t)(r0 + row) * (uint32_t)ld + (uint32_t)(k0 + ls * 8)) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + (it * NTHR + wave * 64) * 16), 16, off, 0, 0, 0);
  }
}
// TR: tile rows = contraction index k0.., columns c0.. of the matrix (128 of them).
template <int BK, int NTHR>
__device__ __forceinline__ void stage_tr(__amdgpu_buffer_rsrc_t rs, char* lds, int k0, int c0, int ld,
                                         int tid, int wave) {
  constexpr int NCH = BK * 16 / NTHR;
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = it * NTHR + tid;
    const int row = c >> 4, pc = c & 15;
    const int lc = pc ^ tr_swz(row);
    const uint32_t off = ((uint32_t)(k0 + row) * (uint32_t)ld + (uint32_t)(c0 + lc * 8)) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + (it * NTHR + wave * 64) * 16), 16, off, 0, 0, 0);
  }
}

// fragment (8 bf16 along k) for the 16 rows/cols [base, base+16) of a staged tile, k-step ks
template <int BK>
__device__ __forceinline__ bf16x8 frag_km(const char* lds, int base, int ks, int lane) {
  const int row = base + (lane & 15), g = lane >> 4;
  const int slot = ks * 4 + g;
  const int addr = row * (BK * 2) + ((slot ^ km_swz<BK>(row)) << 4);
  return *reinterpret_cast<const bf16x8*>(lds + addr);
}
__device__ __forceinline__ bf16x8 frag_tr(const char* lds, int base, int ks, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int col = base + 4 * p;
  const int k1 = ks * 32 + 8 * g + q;
  const int a1 = k1 * 256 + ((((col >> 3)) ^ tr_swz(k1)) << 4) + (col & 7) * 2;
  const int k2 = k1 + 4;
  const int a2 = k2 * 256 + ((((col >> 3)) ^ tr_swz(k2)) << 4) + (col & 7) * 2;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + a1));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + a2));
  typedef short short8v __attribute__((ext_vector_type(8)));
  const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}

}  // namespace gemmc

// host side of the scheduler / launch geometry (gemm.hip)
uint32_t* crl_sched_slot(hipStream_t stream, bool* ok);     // zeroed 16-word state for one persistent launch (rotating pool in device memory owned by the first stream that asks; nullptr = walk statically)
int crl_gemm_cus();             // CUs the persistent kernels spread over: 256 minus crl_gemm_set_reserved_cus
bool crl_gemm_dynamic();        // crl_gemm_set_schedule: dynamic tile tickets (default) or the static walk
