// Fused convolution module of a conformer block for gfx950 (bf16 MFMA, d_model = 256, kernel 9):
//
//     x <- x + W2 silu(dw9(glu(W1 LN(x) + b1)) + bdw) + b2        (model.py:516-544, :670-671)
//
// in ONE kernel; dw9 is the causal depthwise convolution with the eval-mode BatchNorm folded in
// (model.py:236-249).  A workgroup (8 waves) owns 64 frames of one utterance plus the 8 frames in
// front of them that the depthwise taps reach back to (recomputed, not exchanged):
//
//   1. LayerNorm of the 64 + 8 frames into an 80-row bf16 LDS tile (rows 0-7 unused, 8-15 halo).
//   2. GEMM-1: all 512 GLU-interleaved rows of W1 against the 80 frames (20 accumulators per
//      wave); bias + GLU on the accumulators, result written over the LayerNorm tile.
//   3. Depthwise conv + SiLU: every thread slides a 12-row window over 8 channels of 4 frames,
//      reads complete before the 64-row result is written back over the tile.
//   4. GEMM-2: W2 against the 64 frames, then the residual add, as in the fused feed-forward.
//
// The two weight matrices stream through a ring of three 32 KiB LDS stages as one sequence of 12
// uniform slices (direct-to-LDS loads, counted vmcnt, one raw barrier per slice; see gemm.hip);
// the slices of W2 keep arriving while phase 3 runs.  Depthwise weights sit in LDS from the start.
//
// Per-workgroup traffic: 72 x 256 fp32 in, 64 x 256 out, 384 KiB of weights from L2.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int D = 256;            // model width this kernel is built for
constexpr int BM = 64;            // frames per workgroup
constexpr int HALO = 16;          // rows in front of them in the tile (only the last 8 are used)
constexpr int KW = 9;             // depthwise taps
constexpr int NTHREADS = 512;
constexpr int T_BYTES = (BM + HALO) * D * 2;   // 40 KiB frame tile
constexpr int STAGE = 32768;                   // one weight slice: 256 rows x 64 bf16
constexpr int NSTAGE = 3;
constexpr int DW_BYTES = (KW + 1) * D * 4;     // depthwise taps + bias, fp32
constexpr int LDS_BYTES = T_BYTES + NSTAGE * STAGE + DW_BYTES;
constexpr int NSLICE = 12;                     // 8 of W1 (k-slice major, row half minor), 4 of W2

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// [rows][256] bf16 tile (512-byte rows): 16-byte chunk c of row m at c ^ (m & 15)
__device__ __forceinline__ uint32_t wide_off(int row, int chunk) {
  return (uint32_t)row * 512u + (uint32_t)((chunk ^ (row & 15)) << 4);
}

template <int IMM>
__device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct ConvModDev {
  const float* X;           // [M][256] fp32 residual stream
  float* Xout;              // [M][256] updated stream; a different buffer, because a workgroup reads
                            // the 8 frames in front of its own, which its neighbour rewrites
  const float* ln_g;
  const float* ln_b;
  const bf16* W1;           // [512][256], rows interleaved 16 value / 16 gate
  const float* b1;          // [512], same interleave
  const float* dw_w9;       // [9][256] depthwise taps, BatchNorm folded
  const float* dw_b;        // [256]
  const bf16* W2;           // [256][256]
  const float* b2;          // [256]
  int M, Tp;
};

__global__ __launch_bounds__(NTHREADS) void convmod_fused_kernel(const ConvModDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* t_tile = smem;
  char* ring = smem + T_BYTES;
  char* dw_lds = smem + T_BYTES + NSTAGE * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int m0 = blockIdx.x * BM;
  const bool first = (m0 % p.Tp) == 0;        // utterance starts here: the halo is the causal zero padding

  // ---- depthwise taps + bias to LDS (10 KiB, ten 1 KiB wave-instructions over waves 0..7, 0, 1)
  glds16(reinterpret_cast<const char*>(p.dw_w9) + wave * 1024 + lane * 16, dw_lds + wave * 1024);
  if (wave == 0) glds16(reinterpret_cast<const char*>(p.dw_w9) + 8 * 1024 + lane * 16, dw_lds + 8 * 1024);
  if (wave == 1) glds16(reinterpret_cast<const char*>(p.dw_b) + lane * 16, dw_lds + 9 * 1024);

  // ---- weight loader: wave w stages rows 32w..32w+31 of every slice (4 wave-instructions)
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const uint32_t woff_g = (uint32_t)((srow * D + schunk * 8) * 2);
  const char* w1base = reinterpret_cast<const char*>(p.W1) + (size_t)(32 * wave) * D * 2;
  const char* w2base = reinterpret_cast<const char*>(p.W2) + (size_t)(32 * wave) * D * 2;
  constexpr size_t wq = (size_t)8 * D * 2;
  auto issue = [&](int sl_raw) {
    const int sl = sl_raw < NSLICE ? sl_raw : NSLICE - 1;
    char* dst = ring + (sl_raw % NSTAGE) * STAGE + wave * 4096;
    const char* s = sl < 8 ? w1base + ((size_t)(sl & 1) * 256 * D + (sl >> 1) * 64) * 2 : w2base + (size_t)(sl - 8) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(s + q * wq + woff_g, dst + q * 1024);
  };
  issue(0);
  issue(1);      // slice 2 follows from the first gate

  // ---- LayerNorm.  Wave w owns frames 8w..8w+7 in two passes of four: DPP row g (16 lanes) takes
  // frame 8w + 4*pass + g and lane r of the row features 4r + 64k (k = 0..3).  The same mapping is
  // used by the epilogue, so the fp32 input stays in registers for the residual add.  Waves 0 and 1
  // then normalise the 8 halo frames.
  auto layer_norm_row = [&](const f32x4 (&v)[4], int tile_row) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    const float mean = row16_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 c = v[k] - mean;
      q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    }
    const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(p.ln_g + 4 * r + 64 * k);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * r + 64 * k);
      const f32x4 y = (v[k] - mean) * rstd * gg + bb;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
      *reinterpret_cast<bf16x4*>(t_tile + wide_off(tile_row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
    }
  };
  f32x4 rows[2][4], halo[4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int m = m0 + wave * 8 + 4 * ps + g;
#pragma unroll
    for (int k = 0; k < 4; ++k) rows[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
  }
  const bool do_halo = wave < 2 && !first;
  if (do_halo) {
    const int m = m0 - 8 + 4 * wave + g;
#pragma unroll
    for (int k = 0; k < 4; ++k) halo[k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
  }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) layer_norm_row(rows[ps], HALO + wave * 8 + 4 * ps + g);
  if (do_halo) layer_norm_row(halo, 8 + 4 * wave + g);

  // GEMM-1 accumulators: [row half][value / gate][frame tile]; channels 128c + 16w + 4g.., frames 16j + r - 16
  f32x4 acc1[2][2][5];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc1[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bias1[2][2];
  u32x4 wf[2][2], tf[2][5];     // weight fragments by step parity, frame fragments by k32 half

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t woff[2][2], toff[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      woff[a][ks] = lds0 + T_BYTES + (uint32_t)(32 * wave + 16 * a + r) * 128u + (uint32_t)((((ks << 2) | g) ^ (r & 7)) << 4);
      toff[a][ks] = lds0 + (uint32_t)r * 512u + (uint32_t)((((a << 3) | (ks << 2) | g) ^ r) << 4);   // a = k-slice parity
    }

  // gate(sl): slice sl has landed in every wave's part of the ring and every wave is done reading
  // the stage that slice sl+2 will overwrite (and, for sl = 0, the LayerNorm tile is published)
  auto gate = [&](int sl) {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(sl + 2);
  };

  // GEMM-1 step t (0..15): slice t>>1 = (k-slice j, row half c), k32 half t&1.  The frame fragments
  // of a k-slice are read during the c = 0 steps and reused from registers by the c = 1 steps.
  auto load1 = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    constexpr int sl = t >> 1, ks = t & 1, j = sl >> 1, c = sl & 1, buf = t & 1;
    constexpr uint32_t stage = (uint32_t)(sl % NSTAGE) * STAGE;
    lds_read128<0>(wf[buf][0], woff[0][ks] + stage);
    lds_read128<0>(wf[buf][1], woff[1][ks] + stage);
    if constexpr (c == 0) {
      constexpr int timm = (j >> 1) * 256;
      lds_read128<timm>(tf[ks][0], toff[j & 1][ks]);
      lds_read128<timm + 8192>(tf[ks][1], toff[j & 1][ks]);
      lds_read128<timm + 16384>(tf[ks][2], toff[j & 1][ks]);
      lds_read128<timm + 24576>(tf[ks][3], toff[j & 1][ks]);
      lds_read128<timm + 32768>(tf[ks][4], toff[j & 1][ks]);
    }
  };
#define FCENC_CM_FRAGS(buf, ks) "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(tf[ks][0]), "+v"(tf[ks][1]), "+v"(tf[ks][2]), "+v"(tf[ks][3]), "+v"(tf[ks][4])
  auto mma1 = [&](auto tc, auto pc) {
    constexpr int t = decltype(tc)::value, pending = decltype(pc)::value;
    constexpr int sl = t >> 1, ks = t & 1, c = sl & 1, buf = t & 1;
    if constexpr (pending == 7) asm volatile("s_waitcnt lgkmcnt(7)" : FCENC_CM_FRAGS(buf, ks));
    else if constexpr (pending == 6) asm volatile("s_waitcnt lgkmcnt(6)" : FCENC_CM_FRAGS(buf, ks));
    else if constexpr (pending == 2) asm volatile("s_waitcnt lgkmcnt(2)" : FCENC_CM_FRAGS(buf, ks));
    else asm volatile("s_waitcnt lgkmcnt(0)" : FCENC_CM_FRAGS(buf, ks));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        mma(acc1[c][i][j], __builtin_bit_cast(bf16x8, wf[buf][i]), __builtin_bit_cast(bf16x8, tf[ks][j]));
  };

  gate(0);
  load1(std::integral_constant<int, 0>{});
  static_for<0, 16>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (t == 9) {
      // GLU biases (used after step 15) by inline asm, so that the compiler, which cannot count
      // vmcnt across the direct-to-LDS loads, does not drain them; the gate of step 11 retires it
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* bp = p.b1 + 256 * c + 32 * wave + 16 * i + 4 * g;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias1[c][i]) : "v"(bp) : "memory");
        }
    }
    if constexpr (t < 15) {
      constexpr int nt = t + 1;
      if constexpr ((nt & 1) == 0) gate(nt >> 1);
      if constexpr (t == 13)
        asm volatile("" : "+v"(bias1[0][0]), "+v"(bias1[0][1]), "+v"(bias1[1][0]), "+v"(bias1[1][1]));   // valid from here on
      load1(std::integral_constant<int, nt>{});
      mma1(tc, std::integral_constant<int, (((nt >> 1) & 1) == 0) ? 7 : 2>{});
    } else {
      // slice 8 (first of W2) landed and nobody reads the LayerNorm tile any more
      gate(8);
      mma1(tc, std::integral_constant<int, 0>{});
    }
  });
#undef FCENC_CM_FRAGS

  // ---- bias + GLU on the accumulators -> tile[frame row][channel] (model.py:520-521); the halo rows
  // of an utterance's first workgroup are the causal zero padding of the depthwise conv
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ch = 128 * c + 16 * wave + 4 * g;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 v = acc1[c][0][j] + bias1[c][0];
      const f32x4 gt = acc1[c][1][j] + bias1[c][1];
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (bf16)((j == 0 && first) ? 0.f : v[k] * sigmoid_f(gt[k]));
      *reinterpret_cast<bf16x4*>(t_tile + wide_off(16 * j + r, ch >> 3) + 2 * (ch & 7)) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- causal depthwise conv (BatchNorm folded) + SiLU (model.py:526-538): thread = 8 channels x 4 frames
  {
    const int cg = tid & 31, tl = tid >> 5;
    float x[12][8];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(8 + 4 * tl + k, cg));
#pragma unroll
      for (int q = 0; q < 8; ++q) x[k][q] = (float)v[q];
    }
    float y[4][8];
    const float* dwl = reinterpret_cast<const float*>(dw_lds);
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(dwl + KW * D + 8 * cg);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(dwl + KW * D + 8 * cg + 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[s][q] = b0[q];
          y[s][4 + q] = b1[q];
        }
    }
#pragma unroll
    for (int m = 0; m < KW; ++m) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg + 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[s][q] = fmaf(w0[q], x[s + m][q], y[s][q]);
          y[s][4 + q] = fmaf(w1[q], x[s + m][4 + q], y[s][4 + q]);
        }
    }
    // every thread's window reads are complete before any result row overwrites the tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (bf16)silu_f(y[s][q]);
      *reinterpret_cast<bf16x8*>(t_tile + wide_off(4 * tl + s, cg)) = o;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- GEMM-2: slices 8..11 of the ring against tile rows 0..63 ----
  f32x4 acc2[2][4];      // output features 32w + 16i + 4g.., frames 16j + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 hf[2][4];
  auto load2 = [&](auto tc) {
    constexpr int t = decltype(tc)::value;            // 0..7
    constexpr int sl = 8 + (t >> 1), ks = t & 1, j = t >> 1, buf = t & 1;
    constexpr uint32_t stage = (uint32_t)(sl % NSTAGE) * STAGE;
    constexpr int timm = (j >> 1) * 256;
    lds_read128<0>(wf[buf][0], woff[0][ks] + stage);
    lds_read128<0>(wf[buf][1], woff[1][ks] + stage);
    lds_read128<timm>(hf[buf][0], toff[j & 1][ks]);
    lds_read128<timm + 8192>(hf[buf][1], toff[j & 1][ks]);
    lds_read128<timm + 16384>(hf[buf][2], toff[j & 1][ks]);
    lds_read128<timm + 24576>(hf[buf][3], toff[j & 1][ks]);
  };
  auto mma2 = [&](auto tc, auto pc) {
    constexpr int t = decltype(tc)::value, pending = decltype(pc)::value;
    constexpr int buf = t & 1;
#define FCENC_CM_FRAGS2 "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(hf[buf][0]), "+v"(hf[buf][1]), "+v"(hf[buf][2]), "+v"(hf[buf][3])
    if constexpr (pending == 6) asm volatile("s_waitcnt lgkmcnt(6)" : FCENC_CM_FRAGS2);
    else asm volatile("s_waitcnt lgkmcnt(0)" : FCENC_CM_FRAGS2);
#undef FCENC_CM_FRAGS2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mma(acc2[i][j], __builtin_bit_cast(bf16x8, wf[buf][i]), __builtin_bit_cast(bf16x8, hf[buf][j]));
  };
  load2(std::integral_constant<int, 0>{});
  static_for<0, 8>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (t < 7) {
      constexpr int nt = t + 1;
      if constexpr ((nt & 1) == 0) gate(8 + (nt >> 1));
      load2(std::integral_constant<int, nt>{});
      mma2(tc, std::integral_constant<int, 6>{});
    } else {
      mma2(tc, std::integral_constant<int, 0>{});
    }
  });

  // ---- epilogue: out tile [64 frames][256 features] fp32 through LDS (ring area, 64 KiB) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* ct = reinterpret_cast<float*>(ring);
  auto cidx = [](int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * j + r, 32 * wave + 16 * i + 4 * g)) = acc2[i][j];
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g;
    const int m = m0 + row;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r + 64 * k;
      const f32x4 y = rows[ps][k] + *reinterpret_cast<const f32x4*>(ct + cidx(row, col)) +
                      *reinterpret_cast<const f32x4*>(p.b2 + col);
      *reinterpret_cast<f32x4*>(p.Xout + (size_t)m * D + col) = y;
    }
  }
}

}  // namespace

bool convmod_fused_supported(int dtype, int d, int kernel_size) { return dtype == DT_BF16 && d == D && kernel_size == KW; }

int launch_convmod_fused(const ConvModArgs& a, hipStream_t stream) {
  if (a.d != D) return fail("convmod_fused: unsupported width d=%d", a.d);
  if (a.M <= 0 || a.Tp <= 0 || a.Tp % BM || a.M % a.Tp) return fail("convmod_fused: M=%d Tp=%d must be multiples of %d frames", a.M, a.Tp, BM);
  if (a.X == a.Xout) return fail("convmod_fused: the output stream must not alias the input stream");
  if (!a.X || !a.Xout || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.dw_w9 || !a.dw_b || !a.W2 || !a.b2) return fail("convmod_fused: null operand");
  ConvModDev d{};
  d.X = a.X; d.Xout = a.Xout; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const bf16*)a.W1; d.b1 = a.b1; d.dw_w9 = a.dw_w9; d.dw_b = a.dw_b;
  d.W2 = (const bf16*)a.W2; d.b2 = a.b2; d.M = a.M; d.Tp = a.Tp;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(convmod_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          LDS_BYTES) != hipSuccess)
    return fail("convmod_fused: cannot reserve %d bytes of LDS", LDS_BYTES);
  hipLaunchKernelGGL(convmod_fused_kernel, dim3(a.M / BM), dim3(NTHREADS), LDS_BYTES, stream, d);
  return check_launch("convmod_fused");
}

}  // namespace fcenc
