// Shared device helpers for the FastConformer encoder kernels (gfx950 / CDNA4 only).
//
// Conventions used by every MFMA kernel in this directory:
//   * wave = 64 lanes; lane = 16*g + r with g = lane>>4 (0..3), r = lane&15.
//   * one "k32 step" contracts 32 elements of K.  A fragment holds, for lane (g, r), the 8
//     consecutive K elements 8g..8g+7 of row r of its operand tile.
//       bf16 : one v_mfma_f32_16x16x32_bf16 per k32 step (8 bf16 = 4 VGPRs per operand)
//       fp32 : eight v_mfma_f32_16x16x4_f32 per k32 step; MFMA j takes element j of both
//              fragments (K slot (g, j) of the instruction = K element 8g+j of the tile), which
//              is a bijection of K shared by both operands, so the sum is the same.
//   * accumulator (both dtypes): lane (g, r), register q holds D[row = 4g+q][col = r].
//   * LDS tiles are rows of ROWB bytes cut into 16-byte chunks, chunk c of row w stored at
//     chunk c ^ (w & 7): conflict-free for ds_read_b128 fragment reads (16 rows x one chunk),
//     for the 8-lane-per-row staging writes and for ds_read_b64_tr_b16 blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcenc {

using bf16 = __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

struct f32x8 {
  f32x4 lo, hi;
};

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };

template <typename T>
struct Traits;
template <>
struct Traits<bf16> {
  using Frag = bf16x8;
  using Vec4 = bf16x4;                // four consecutive outputs of one lane
  static constexpr int EPC = 8;       // elements per 16-byte chunk
};
template <>
struct Traits<float> {
  using Frag = f32x8;
  using Vec4 = f32x4;
  static constexpr int EPC = 4;
};

__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#ifdef FCENC_ABL_NOMFMA
  // timing experiment (wrong results): one VALU instruction that keeps both operands alive instead of the MFMA
  acc[0] = fmaf(__builtin_bit_cast(f32x4, a)[0], __builtin_bit_cast(f32x4, b)[0], acc[0]);
#else
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}

// Phase clock for tools/diag_block.py (builds with -DFCENC_STAMPS only; the shipped library compiles it away):
// lane 0 of every wave leaves the shader clock in slot i (ignored when negative) of its row of a table at a fixed LDS offset behind every
// kernel's own tiles, and the block kernel copies the table out when it ends.
#ifdef FCENC_STAMPS
constexpr int STAMP_OFF = 150 * 1024, STAMP_SLOTS = 64, STAMP_BYTES = 8 * STAMP_SLOTS * 8;
#define FC_STAMP(smem_, i)                                                                                   \
  do {                                                                                                       \
    if ((i) >= 0 && (threadIdx.x & 63) == 0)                                                                 \
      reinterpret_cast<unsigned long long*>((smem_) + fcenc::STAMP_OFF)[(threadIdx.x >> 6) * fcenc::STAMP_SLOTS + (i)] = \
          __builtin_amdgcn_s_memtime();                                                                      \
  } while (0)
#else
#define FC_STAMP(smem_, i) do {} while (0)
#endif
__device__ __forceinline__ void mma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
}

// byte offset of 16-byte chunk `chunk` of row `row` in a swizzled tile with ROWB-byte rows
template <int ROWB>
__device__ __forceinline__ uint32_t tile_off(int row, int chunk) {
  return (uint32_t)row * ROWB + (uint32_t)((chunk ^ (row & 7)) << 4);
}

// fragment of row `row`, k32 step `ks`, for lane group g
template <typename T, int ROWB>
__device__ __forceinline__ typename Traits<T>::Frag lds_frag(const char* tile, int row, int ks, int g);
template <>
__device__ __forceinline__ bf16x8 lds_frag<bf16, 128>(const char* tile, int row, int ks, int g) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off<128>(row, ks * 4 + g));
}
template <>
__device__ __forceinline__ f32x8 lds_frag<float, 128>(const char* tile, int row, int ks, int g) {
  f32x8 f;
  f.lo = *reinterpret_cast<const f32x4*>(tile + tile_off<128>(row, ks * 8 + 2 * g));
  f.hi = *reinterpret_cast<const f32x4*>(tile + tile_off<128>(row, ks * 8 + 2 * g + 1));
  return f;
}
template <>
__device__ __forceinline__ f32x8 lds_frag<float, 256>(const char* tile, int row, int ks, int g) {
  f32x8 f;
  f.lo = *reinterpret_cast<const f32x4*>(tile + tile_off<256>(row, ks * 8 + 2 * g));
  f.hi = *reinterpret_cast<const f32x4*>(tile + tile_off<256>(row, ks * 8 + 2 * g + 1));
  return f;
}

// v_exp + v_rcp (1 ulp each): a handful of VALU instructions instead of an IEEE division sequence
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// SiLU of x given u = x log2(e): x / (1 + e^-x) = u / (log2e + log2e 2^-u).  One multiply fewer per value than
// silu_f when the caller gets u for free (weights and bias pre-scaled by log2e, or a bias add turned into an FMA).
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float silu_log2e(float u) {
  return u * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(-u), kLog2e, kLog2e));
}

// sum over the 16 lanes of a DPP row (lanes 16k..16k+15), result in every lane, VALU only:
// quad butterflies, then the two mirrors swap quads and halves whose lanes already agree
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// Reductions over the four DPP rows of a wave (lanes r, r + 16, r + 32, r + 48), result in all four: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange whole rows / wave halves between two registers in one VALU
// instruction each (x copied to both: {rows 0,0,2,2} and {rows 1,1,3,3}, then {half 0, half 0} and {half 1, half 1}),
// where __shfl_xor(x, 16 / 32) takes two dependent trips through the LDS crossbar (ds_bpermute).  Same operands in
// the same association as x += shfl_xor(x, 16); x += shfl_xor(x, 32), so sums are bit-identical.
// Both results pass through an empty asm statement: without it this compiler (ROCm 7.2 hipcc) treats the two results
// of a swap as one value and folds max(a, b) / a + b of them into max(a, a) / a + a (seen in the ISA; a probe
// kernel that stores both results shows them different, as the ISA manual says).
__device__ __forceinline__ void swap_rows16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  unsigned p0 = p[0], p1 = p[1];
  asm volatile("" : "+v"(p0), "+v"(p1));
  a = __builtin_bit_cast(float, p0);
  b = __builtin_bit_cast(float, p1);
}
__device__ __forceinline__ void swap_halves32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  unsigned p0 = p[0], p1 = p[1];
  asm volatile("" : "+v"(p0), "+v"(p1));
  a = __builtin_bit_cast(float, p0);
  b = __builtin_bit_cast(float, p1);
}
// v_permlane16_swap on two different registers: afterwards a = {a.row0, b.row0, a.row2, b.row2} and
// b = {a.row1, b.row1, a.row3, b.row3} (rows of 16 lanes)
__device__ __forceinline__ void swap_odd_even_rows(unsigned& a, unsigned& b) {
  const auto p = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  unsigned p0 = p[0], p1 = p[1];
  asm volatile("" : "+v"(p0), "+v"(p1));
  a = p0;
  b = p1;
}
__device__ __forceinline__ float rows4_max(float v) {
  float a, b;
  swap_rows16(v, a, b);
  swap_halves32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float rows4_sum(float v) {
  float a, b;
  swap_rows16(v, a, b);
  swap_halves32(a + b, a, b);
  return a + b;
}

__device__ __forceinline__ void store4(bf16* p, const f32x4& v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
  *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// Stores through a buffer descriptor: the hardware drops what lies beyond `bytes` (and everything when the pointer
// is null), so a ragged last tile or an optional output costs no branch.  That matters beyond the branch itself:
// the compiler counts outstanding memory operations per straight-line path, and a store it cannot count makes the
// next wait for an older load wait for every store as well.  Buffers are below 2 GiB; an offset of BUF_SKIP or more
// is out of range for all of them.
constexpr uint32_t BUF_SKIP = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
// AUX: cache policy bits of the store (0 = plain write-back, 16 = sc1 write-through, 2 = nt; measured on the block
// kernel's outputs: write-through no faster, nt and write-through of 32-byte pieces slower)
template <int AUX = 0>
__device__ __forceinline__ void buffer_store4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, byte_off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void buffer_store4_bf16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, const f32x4& v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsrc, byte_off, 0, AUX);
}

template <int AUX = 0>
__device__ __forceinline__ void buffer_store8_bf16(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, const f32x4& lo,
                                                   const f32x4& hi) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = (bf16)lo[i];
    o[4 + i] = (bf16)hi[i];
  }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrc, byte_off, 0, AUX);
}

__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }

// eight consecutive elements as fp32
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <>
__device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = a[i];
    o[4 + i] = b[i];
  }
}

// XCD-aware bijective remap of a linear block id: blocks id, id+8, id+16, ... (which share an
// XCD under round-robin placement) receive consecutive logical tiles, so tiles that share
// operand panels hit the same L2.  Placement is only a speed assumption, never correctness.
__device__ __forceinline__ int xcd_remap(int id, int nblocks) {
  const int q = nblocks >> 3, rem = nblocks & 7, x = id & 7, s = id >> 3;
  return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + s;
}

}  // namespace fcenc
