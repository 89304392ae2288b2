// Fragment access for 64-row x 64-bf16 LDS tiles (128-byte rows, one XOR swizzle that is conflict-free for both the ds_read_b128 row
// reads and the ds_read_b64_tr_b16 transposed reads) and the v_mfma_f32_32x32x16_bf16 accumulator helpers shared by the flash
// attention kernels (attention.hip) and the Swin window attention kernels (swin.hip).
#pragma once
#include "common.h"

namespace attnf {

__device__ __forceinline__ int swz64(int row) {
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
}

// Per-lane LDS byte offsets (tile base 0) of every fragment read, computed ONCE before the tile loop and made
// opaque so hipcc keeps them in registers instead of re-deriving ~100 integer ops per tile; tile / block bases are
// compile-time constants that fold into the ds_read offset field.
struct LaneAddr {
  uint32_t row[4];     // frag_row: [ks], rows 0..31 (+4096 B for rows 32..63)
  uint32_t tr[2][2];   // frag_tr: [db][first / second 4-row group], kbase 0 (+128*kbase B)
};
__device__ __forceinline__ LaneAddr make_lane_addr(int lane) {
  LaneAddr la;
  const int row = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) la.row[ks] = (uint32_t)(row * 128 + (((2 * ks + hh) ^ swz64(row)) << 4));
  const int G = lane >> 4, idx = lane & 15, qq = idx >> 2, p = idx & 3;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int col = 32 * db + 16 * (G & 1) + 4 * p;
    const int r1 = 4 * hh + qq, r2 = r1 + 8;
    la.tr[db][0] = (uint32_t)(r1 * 128 + (((col >> 3) ^ swz64(r1)) << 4) + (col & 7) * 2);
    la.tr[db][1] = (uint32_t)(r2 * 128 + (((col >> 3) ^ swz64(r2)) << 4) + (col & 7) * 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(la.row[i]));
  asm volatile("" : "+v"(la.tr[0][0]), "+v"(la.tr[0][1]), "+v"(la.tr[1][0]), "+v"(la.tr[1][1]));
  return la;
}
// MFMA 32x32x16 operand whose k index is contiguous in the tile row: rows rowbase..+31, k = 16ks + 8h + j
__device__ __forceinline__ bf16x8 frag_row(const char* tile, const LaneAddr& la, int rowbase, int ks) {
  return *reinterpret_cast<const bf16x8*>(tile + la.row[ks] + rowbase * 128);
}
// MFMA 32x32x16 operand whose k index is the tile ROW: operand row = tile column 32db + (lane&31),
// element j of lane half h = tile row kbase + 8(j>>2) + 4h + (j&3)  (the accumulator k order)
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, const LaneAddr& la, int kbase, int db) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + la.tr[db][0] + kbase * 128));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + la.tr[db][1] + kbase * 128));
  typedef short short8v __attribute__((ext_vector_type(8)));
  const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// value held by lane ^ 32
__device__ __forceinline__ float swap32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// registers 8s..8s+7 of a 32x32 accumulator as the next product's operand
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
  return r;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// c += a.b with the accumulator TIED to the destination (asm: with 128 long-lived accumulator registers hipcc allocates the builtin's
// destination apart from its C operand and spills).  hipcc does not see an MFMA here, so it inserts none of the wait states an
// operand written by the preceding VALU instruction (v_cvt_pk_bf16_f32) may need: the s_nop in front supplies them; the accumulators
// themselves are only read by further MFMAs (interlocked in hardware) until the kernel's epilogue, which settles the pipe first.
__device__ __forceinline__ void mfma32_acc(bf16x8 a, bf16x8 b, f32x16& c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// the same with the accumulator in the ACCUMULATOR half of the register file (a kernel with one wave per SIMD owns 256 + 256 registers)
__device__ __forceinline__ void mfma32_acc_a(bf16x8 a, bf16x8 b, f32x16& c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// row index (within a 32-row block) of accumulator register r for lane half hh
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

}  // namespace attnf
