// Fused convolution module of a conformer block for gfx950 (bf16 MFMA, d_model = 256, kernel 9):
//
//     xout = x + W2 silu(dw9(glu(W1 LN(x) + b1)) + bdw) + b2        (model.py:516-544, :670-671)
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
//   4. GEMM-2: W2 against the 64 frames, then the residual add.
//
// Weights never touch LDS: wave w multiplies only rows 32w..32w+31 of each 256-row block, so it
// streams those rows from L2 straight into MFMA A-fragments through a ring of registers, four
// steps (16 KiB per wave) ahead of their use; all of W2 is already in flight when phase 2 ends.
// The weights are stored in that streaming order (convmod_pack below).  See ffn.hip for why.
//
// Per-workgroup traffic: 72 x 256 fp32 in, 64 x 256 out, 384 KiB of weights from L2.
#include <type_traits>

#include "common.h"
#include "ffn_body.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int D = 256;            // model width this kernel is built for
constexpr int BM = 64;            // frames per workgroup
constexpr int HALO = 16;          // rows in front of them in the tile (only the last 8 are used)
constexpr int KW = 9;             // depthwise taps
constexpr int NTHREADS = 512;
constexpr int T_BYTES = (BM + HALO) * D * 2;   // 40 KiB frame tile
constexpr int CT_BYTES = BM * D * 4;           // 64 KiB fp32 output tile of the epilogue, over the frame tile
constexpr int DW_BYTES = (KW + 1) * D * 4;     // depthwise taps + bias, fp32
constexpr int DEPTH = 4;                       // weight steps in flight per wave (4 fragments each)
constexpr int NSTEP = 12;                      // 8 k32 steps of GEMM-1, 4 double k32 steps of GEMM-2

// [rows][256] bf16 tile (512-byte rows): 16-byte chunk c of row m at c ^ (m & 15)
__device__ __forceinline__ uint32_t wide_off(int row, int chunk) {
  return (uint32_t)row * 512u + (uint32_t)((chunk ^ (row & 15)) << 4);
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#ifdef FCENC_STAMPS
constexpr int STAMP_MAX_WG = 1024;
__device__ unsigned long long g_stamps[STAMP_MAX_WG * 512];     // [workgroup][wave][slot] of the last chained launch
#endif

struct ConvModDev {
  const float* X;           // [M][256] fp32 residual stream
  float* Xout;              // [M][256] updated stream; a different buffer, because a workgroup reads
                            // the 8 frames in front of its own, which its neighbour rewrites
  const float* ln_g;
  const float* ln_b;
  const u32x4* W1;          // packed, see convmod_pack_kernel
  const float* b1;          // [512], rows interleaved 16 value / 16 gate
  const float* dw_w9;       // [9][256] depthwise taps, BatchNorm folded
  const float* dw_b;        // [256]
  const u32x4* W2;          // packed
  const float* b2;          // [256]
  // OPROJ only: the stream entering the module is X + O Wo^T + bo (attention out_proj + residual)
  const bf16* O;            // [M][256] attention output
  const u32x4* Wo;          // [256][256], row-packed (rows_pack_kernel of ffn.hip)
  const float* bo;          // [256]
  int M, Tp;
  // FFN > 0 only: the feed-forward block that follows the module (its output X is this Xout)
  ffnk::FfnDev ffn;
  // CHAIN only: the NEXT conformer block's first feed-forward module and q/k/v projection, run on the
  // rows the closing LayerNorm has just produced (its X is this Xout as well)
  ffnk::FfnDev next;
};

// LDS map.  Plain: frame tile at 0 (the epilogue's fp32 tile reuses it), depthwise weights behind.
// OPROJ: an 80 KiB region first (attention-output tile, then the fp32 projection of all 80 rows,
// later the epilogue's tile), the frame tile and the depthwise weights behind it.
// Last, the module's parameter vectors (LayerNorm affine, out_proj bias, both pointwise biases; 6 KiB): read from
// LDS they do not queue behind the weights in flight, as a global load in the middle of the kernel would.
constexpr int CP_LN_G = 0, CP_LN_B = 256, CP_BO = 512, CP_B2 = 768, CP_B1 = 1024, CP_FLOATS = 1536;
template <bool OPROJ>
struct Lds {
  static constexpr int A_BYTES = OPROJ ? (BM + HALO) * D * 4 : 0;
  static constexpr int T_OFF = A_BYTES;
  static constexpr int DW_OFF = OPROJ ? A_BYTES + T_BYTES : CT_BYTES;
  static constexpr int PRM_OFF = DW_OFF + DW_BYTES;
  static constexpr int BYTES = PRM_OFF + CP_FLOATS * 4;
  // behind them the parameter areas of the feed-forward blocks a kernel goes on to run (FFN > 0, CHAIN)
  static constexpr int FFN_PRM_OFF = BYTES;
  static constexpr int NEXT_PRM_OFF = FFN_PRM_OFF + ffnk::Prm<0>::BYTES;
  static constexpr int bytes(int ffn, bool chain) {
    return ffn == 0 ? BYTES : chain ? NEXT_PRM_OFF + ffnk::Prm<3>::BYTES : NEXT_PRM_OFF;
  }
};

// FFN = number of 256-unit hidden chunks of the feed-forward block run on the module's output by the
// same workgroup (0: none; the module's output then goes to Xout).
// CHAIN: after that, the next block's first feed-forward module + q/k/v projection (FFN chunks again).
template <bool OPROJ, int FFN, bool CHAIN = false>
__global__ __launch_bounds__(NTHREADS) void convmod_fused_kernel(const ConvModDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(FFN == 0 || Lds<OPROJ>::DW_OFF >= ffnk::LDS_BYTES,
                "the feed-forward body's tiles (from offset 0) must end before the areas that live through it");
  char* t_tile = smem + Lds<OPROJ>::T_OFF;
  float* dwl = reinterpret_cast<float*>(smem + Lds<OPROJ>::DW_OFF);
  float* prm = reinterpret_cast<float*>(smem + Lds<OPROJ>::PRM_OFF);
  auto cidx = [](int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FCENC_PRIO_YOUNG
  // experiment: the second-dispatched wave of each SIMD loses the issue arbitration to the older one in every
  // phase; a static priority for that half (the condition is wave-uniform: s_setprio ignores EXEC)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int g = lane >> 4, r = lane & 15;
  const int m0 = blockIdx.x * BM;
  const bool first = (m0 % p.Tp) == 0;        // utterance starts here: the halo is the causal zero padding

  // ---- weight stream: step s reads four consecutive 1 KiB fragments of this wave's run ----
  constexpr int S0 = OPROJ ? 4 : 0;           // steps of Wo in front of the 8 of W1 and the 4 of W2
  constexpr int NS = S0 + NSTEP;
  const u32x4* wos = p.Wo + (size_t)wave * 1024 + lane;     // 4 steps x [k32 half][fragment i]
  const u32x4* w1s = p.W1 + (size_t)wave * 2048 + lane;     // 8 steps x [half c][fragment i]
  const u32x4* w2s = p.W2 + (size_t)wave * 1024 + lane;     // 4 steps x [k32 half][fragment i]
  u32x4 wr[DEPTH][4];
  auto fetch = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
#ifdef FCENC_ABL_NOLOAD
    if constexpr (s >= DEPTH) return;      // timing experiment (wrong results): the ring is filled once and never again
#endif
    if constexpr (s < NS) {
      const u32x4* src = s < S0 ? wos + s * 256 : s < S0 + 8 ? w1s + (s - S0) * 256 : w2s + (s - S0 - 8) * 256;
#pragma unroll
      for (int f = 0; f < 4; ++f) wr[s % DEPTH][f] = src[64 * f];
    }
  };
#ifdef FCENC_STAMPS
  reinterpret_cast<unsigned long long*>(smem + STAMP_OFF)[tid] = 0;
  __syncthreads();
#endif
  FC_STAMP(smem, 0);

  // Loads return in the order they were issued, so what the first phase needs is issued first: the attention
  // output tile (OPROJ), the depthwise taps and the parameter vectors, then the first ring of weights.
  u32x4 ov[OPROJ ? 5 : 1];
  if constexpr (OPROJ) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + NTHREADS * i;
      const int row = c >> 5, ch = c & 31;
      ov[i] = u32x4{0u, 0u, 0u, 0u};
      if (row >= 8 && (row >= HALO || !first))
        ov[i] = *reinterpret_cast<const u32x4*>(p.O + ((size_t)m0 + row - HALO) * D + 8 * ch);
    }
  }
  // ---- depthwise taps + bias to LDS (10 KiB) ----
  const f32x4 taps_a = *reinterpret_cast<const f32x4*>(p.dw_w9 + 4 * tid);            // taps 0..7
  f32x4 taps_b = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 64) taps_b = *reinterpret_cast<const f32x4*>(p.dw_w9 + 8 * D + 4 * tid);  // tap 8
  else if (tid < 128) taps_b = *reinterpret_cast<const f32x4*>(p.dw_b + 4 * (tid - 64));
  f32x4 prm_v = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int f = 4 * tid;                      // segments are multiples of 256 floats: uniform per wave
    const float* src = f < CP_LN_B ? p.ln_g + f : f < CP_BO ? p.ln_b + (f - CP_LN_B) : f < CP_B2 ? (OPROJ ? p.bo + (f - CP_BO) : nullptr)
                       : f < CP_B1 ? p.b2 + (f - CP_B2) : f < CP_FLOATS ? p.b1 + (f - CP_B1) : nullptr;
    if (src) prm_v = *reinterpret_cast<const f32x4*>(src);
  }
  ffnk::PrmStage<0> ffn_prm;
  ffnk::PrmStage<3> next_prm;
  if constexpr (FFN > 0) ffn_prm.issue(p.ffn);
  if constexpr (CHAIN) next_prm.issue(p.next);
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
  __builtin_amdgcn_sched_barrier(0);
  f32x4 xin[2][4], xhalo[4];
  const bool do_halo = wave < 2 && !first;
  auto load_rows = [&]() {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) xin[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
    }
    if (do_halo) {
      const int m = m0 - 8 + 4 * wave + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) xhalo[k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
    }
  };
  *reinterpret_cast<f32x4*>(dwl + 4 * tid) = taps_a;
  if (tid < 128) *reinterpret_cast<f32x4*>(dwl + 8 * D + 4 * tid) = taps_b;
  if (4 * tid < CP_FLOATS) *reinterpret_cast<f32x4*>(prm + 4 * tid) = prm_v;
  if constexpr (FFN > 0) ffn_prm.commit(reinterpret_cast<float*>(smem + Lds<OPROJ>::FFN_PRM_OFF));
  if constexpr (CHAIN) next_prm.commit(reinterpret_cast<float*>(smem + Lds<OPROJ>::NEXT_PRM_OFF));
  if constexpr (!OPROJ) __syncthreads();      // (with OPROJ the barrier behind the attention-output tile publishes them)

  // ---- LayerNorm.  Wave w owns frames 8w..8w+7 in two passes of four: DPP row g (16 lanes) takes
  // frame 8w + 4*pass + g and lane r of the row features 4r + 64k (k = 0..3); the epilogue uses the
  // same mapping.  Waves 0 and 1 then normalise the 8 halo frames.
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
      const f32x4 gg = *reinterpret_cast<const f32x4*>(prm + CP_LN_G + 4 * r + 64 * k);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(prm + CP_LN_B + 4 * r + 64 * k);
      const f32x4 y = (v[k] - mean) * rstd * gg + bb;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
      *reinterpret_cast<bf16x4*>(t_tile + wide_off(tile_row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
    }
  };
  if constexpr (OPROJ) {
    // ---- attention out_proj for the 72 frames (model.py:429): O tile -> LDS, W against it, fp32
    // result transposed through LDS so that the LayerNorm below can add it row by row ----
    char* o_tile = smem;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + NTHREADS * i;
      *reinterpret_cast<u32x4*>(o_tile + wide_off(c >> 5, c & 31)) = ov[i];
    }
    __syncthreads();
    FC_STAMP(smem, 1);
    // the fp32 rows of the stream are wanted when the projection is done: requested here, behind the loads the
    // projection itself waits for (a wave cannot pass a load it cannot issue, and the CU issues 64 bytes a clock)
    load_rows();
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0[2][5];      // features 32w + 16i + 4g.., frames 16j + r - 16
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 of[2][5];
    auto frames0 = [&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks < 8) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
          of[ks & 1][j] = *reinterpret_cast<const bf16x8*>(o_tile + wide_off(16 * j + r, 4 * ks + g));
      }
    };
    frames0(std::integral_constant<int, 0>{});
    static_for<0, 8>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int s = ks >> 1;
      __builtin_amdgcn_sched_barrier(0);
      frames0(std::integral_constant<int, ks + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
          mma(acc0[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][2 * (ks & 1) + i]), of[ks & 1][j]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks & 1) fetch(std::integral_constant<int, s + DEPTH>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    FC_STAMP(smem, 2);
    __syncthreads();       // the O tile has been read; its space now takes the projection
    float* ct0 = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        *reinterpret_cast<f32x4*>(ct0 + cidx(16 * j + r, 32 * wave + 16 * i + 4 * g)) = acc0[i][j];
    __syncthreads();
    FC_STAMP(smem, 3);
  }
  {
    if constexpr (!OPROJ) load_rows();
    if constexpr (OPROJ) {
      // x + out_proj(attention) (model.py:667); the 64 own rows are parked where their projection stood, in the
      // fp32 tile (nothing else uses that region until the epilogue, which adds the module's output to them in
      // place): holding them would cost 32 registers through the phases in between, a round trip through global
      // memory 128 KiB of the CU's store and load paths
      float* ct0 = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * r + 64 * k;
        const f32x4 bo = *reinterpret_cast<const f32x4*>(prm + CP_BO + col);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int row = wave * 8 + 4 * ps + g;
          xin[ps][k] += *reinterpret_cast<const f32x4*>(ct0 + cidx(HALO + row, col)) + bo;
          *reinterpret_cast<f32x4*>(ct0 + cidx(HALO + row, col)) = xin[ps][k];
        }
        if (do_halo) xhalo[k] += *reinterpret_cast<const f32x4*>(ct0 + cidx(8 + 4 * wave + g, col)) + bo;
      }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) layer_norm_row(xin[ps], HALO + wave * 8 + 4 * ps + g);
    if (do_halo) layer_norm_row(xhalo, 8 + 4 * wave + g);
  }
  __syncthreads();
  FC_STAMP(smem, 4);

  // ---- GEMM-1.  acc1[half c][value / gate][frame tile]: channels 128c + 16w + 4g.., frames 16j + r - 16
  f32x4 acc1[2][2][5];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc1[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One step = this step's five frame fragments (LDS; the MFMAs start as the first one lands, and
  // a second set of fragment registers would not fit beside 80 accumulators and the weight ring),
  // its 20 MFMAs, then the weight fetch DEPTH steps ahead into the registers just released.  The
  // scheduling fences keep the compiler from sinking the fetches towards their use; it still places
  // the counted waits itself.
  static_for<0, 8>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 tf[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
      tf[j] = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(16 * j + r, 4 * s + g));
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          mma(acc1[c][i][j], __builtin_bit_cast(bf16x8, wr[(S0 + s) % DEPTH][2 * c + i]), tf[j]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, S0 + s + DEPTH>{});
    __builtin_amdgcn_sched_barrier(0);
  });
  FC_STAMP(smem, 5);
  __syncthreads();      // nobody reads the LayerNorm tile any more

  // ---- bias + GLU on the accumulators -> tile[frame row][channel] (model.py:520-521); the halo rows
  // of an utterance's first workgroup are the causal zero padding of the depthwise conv
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ch = 128 * c + 16 * wave + 4 * g;
    const f32x4 bias_v = *reinterpret_cast<const f32x4*>(prm + CP_B1 + 256 * c + 32 * wave + 4 * g);
    const f32x4 bias_g = *reinterpret_cast<const f32x4*>(prm + CP_B1 + 256 * c + 32 * wave + 16 + 4 * g);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 v = acc1[c][0][j] + bias_v;
      const f32x4 gt = acc1[c][1][j] + bias_g;
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (bf16)((j == 0 && first) ? 0.f : v[k] * sigmoid_f(gt[k]));
      *reinterpret_cast<bf16x4*>(t_tile + wide_off(16 * j + r, ch >> 3) + 2 * (ch & 7)) = o;
    }
  }
  __syncthreads();
  FC_STAMP(smem, 6);

  // ---- causal depthwise conv (BatchNorm folded) + SiLU (model.py:526-538): thread = 8 channels x 4 frames
  {
    const int cg = tid & 31, tl = tid >> 5;
    // taps and outputs stay in registers, the 12 window rows pass through one at a time (the W2
    // fragments prefetched for GEMM-2 occupy 64 registers throughout)
    float w[KW][8], y[4][8];
#pragma unroll
    for (int m = 0; m <= KW; ++m) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (m < KW) {
          w[m][q] = w0[q];
          w[m][4 + q] = w1[q];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            y[s][q] = w0[q];
            y[s][4 + q] = w1[q];
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(8 + 4 * tl + k, cg));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int m = k - s;                      // frame s takes row k through tap m
        if (m < 0 || m >= KW) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) y[s][q] = fmaf(w[m][q], (float)v[q], y[s][q]);
      }
    }
    __syncthreads();    // every thread's window reads are complete before any result row overwrites the tile
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (bf16)silu_f(y[s][q]);
      *reinterpret_cast<bf16x8*>(t_tile + wide_off(4 * tl + s, cg)) = o;
    }
  }
  __syncthreads();
  FC_STAMP(smem, 7);

  // Lane coordinates re-derived behind an optimisation barrier: otherwise the compiler keeps the
  // addresses it formed from them in the first phases alive, and spilled, across the whole kernel.
  int lane_b = lane;
  asm volatile("" : "+v"(lane_b));
  const int g2 = lane_b >> 4, r2 = lane_b & 15;

  // the fp32 input rows again for the residual add (L2 hits; holding them since the LayerNorm would cost 32
  // registers through the two register-hungry phases above); with OPROJ they wait in the fp32 tile instead
  f32x4 rows[2][4];
  if constexpr (!OPROJ) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + g2;
#pragma unroll
      for (int k = 0; k < 4; ++k) rows[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r2 + 64 * k);
    }
  }

  // ---- GEMM-2: W2 (already in registers or in flight) against tile rows 0..63 ----
  f32x4 acc2[2][4];      // output features 32w + 16i + 4g.., frames 16j + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 hf[2][4];
  auto frames2 = [&](auto kc) {
    constexpr int ks = decltype(kc)::value;        // k32 step 0..7
    if constexpr (ks < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hf[ks & 1][j] = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(16 * j + r2, 4 * ks + g2));
    }
  };
  frames2(std::integral_constant<int, 0>{});
  static_for<0, 8>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    constexpr int s = S0 + 8 + (ks >> 1);
    __builtin_amdgcn_sched_barrier(0);
    frames2(std::integral_constant<int, ks + 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mma(acc2[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][2 * (ks & 1) + i]), hf[ks & 1][j]);
    __builtin_amdgcn_sched_barrier(0);
  });

  // ---- epilogue: out tile [64 frames][256 features] fp32 through LDS (over the frame tile) ----
  FC_STAMP(smem, 8);
  __syncthreads();
  // OPROJ: the tile's rows HALO.. hold x + out_proj; the module's output is added to them where they lie
  float* ct = reinterpret_cast<float*>(smem) + (OPROJ ? HALO * D : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4* at = reinterpret_cast<f32x4*>(ct + cidx(16 * j + r2, 32 * wave + 16 * i + 4 * g2));
      if constexpr (OPROJ) *at += acc2[i][j] + *reinterpret_cast<const f32x4*>(prm + CP_B2 + 32 * wave + 16 * i + 4 * g2);
      else *at = acc2[i][j];
    }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g2;
    const int m = m0 + row;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r2 + 64 * k;
      const f32x4 out = *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
      rows[ps][k] = OPROJ ? out : rows[ps][k] + (out + *reinterpret_cast<const f32x4*>(prm + CP_B2 + col));
      if constexpr (FFN == 0) *reinterpret_cast<f32x4*>(p.Xout + (size_t)m * D + col) = rows[ps][k];
    }
  }
  FC_STAMP(smem, 9);
  if constexpr (FFN > 0) {
    // x += 0.5 * ff2(x), x = LayerNorm(x) (model.py:674, :677) on the rows still in registers
    __syncthreads();        // the fp32 tile has been read; the feed-forward tiles take its place
    auto hand_over = [&](f32x4 (&in)[2][4]) {
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int k = 0; k < 4; ++k) in[ps][k] = rows[ps][k];
    };
    float* ffn_area = reinterpret_cast<float*>(smem + Lds<OPROJ>::FFN_PRM_OFF);
    auto no_emit = [](f32x4 (&)[2][4]) {};
    if constexpr (!CHAIN) {
      ffnk::ffn_block<FFN, 0, true, decltype(hand_over)&, decltype(no_emit)&, false, true>(p.ffn, smem, m0, hand_over, no_emit,
                                                                                          16, ffn_area);
    } else {
      // the block's output is only an intermediate here: it stays in registers and the next block's
      // x += 0.5 * ff1(x) (model.py:663) and q | k | v projection follow at once
      auto keep = [&](f32x4 (&y)[2][4]) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int k = 0; k < 4; ++k) rows[ps][k] = y[ps][k];
      };
      ffnk::ffn_block<FFN, 0, false, decltype(hand_over)&, decltype(keep)&, false, true>(p.ffn, smem, m0, hand_over, keep, 16,
                                                                                        ffn_area);
      __syncthreads();      // the fp32 tile of the block above has been read
      ffnk::ffn_block<FFN, 3, true, decltype(hand_over)&, decltype(no_emit)&, false, true>(
          p.next, smem, m0, hand_over, no_emit, 32, reinterpret_cast<float*>(smem + Lds<OPROJ>::NEXT_PRM_OFF));
    }
  }
#ifdef FCENC_STAMPS
  FC_STAMP(smem, 48);
  __syncthreads();
  if (CHAIN && blockIdx.x < STAMP_MAX_WG)
    g_stamps[blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem + STAMP_OFF)[threadIdx.x];
#endif
}


// ---------------------------------------------------------------------------------------------
// The same module for the cached streaming step (model.py:835-894 with cache; conv cache model.py:236-249):
//
//     x <- x + out_proj(attention) ;  x <- x + W2 silu(dw9([cache | glu(W1 LN(x) + b1)]) + bdw) + b2
//
// on tokens packed stream-major (row = stream * Tq + i, Tq <= 16 frames of this chunk per stream).  A workgroup
// owns NS = 64 / Tq whole streams (NS * Tq <= 64 rows), so the depthwise taps never cross a workgroup: a frame's
// history is its stream's earlier rows of the tile and, before those, the stream's conv cache [8][d] (the last 8 GLU
// frames of the previous chunks), which the kernel then replaces by the last 8 frames of [cache | chunk].  Rows are
// only ever touched by their own workgroup, so X is updated in place.  Weight streaming, packing and the GEMM
// phases are those of convmod_fused_kernel<true, 0> without the halo rows.
// ---------------------------------------------------------------------------------------------
struct StreamConvModDev {
  float* X;                 // [B * Tq][256] fp32 residual stream, updated in place
  const bf16* O;            // [B * Tq][256] attention output
  const u32x4* Wo;          // row-packed
  const float* bo;
  const float* ln_g;
  const float* ln_b;
  const u32x4* W1;          // convmod_pack'ed
  const float* b1;
  const float* dw_w9;
  const float* dw_b;
  const u32x4* W2;
  const float* b2;
  bf16* cache;              // [B][8][256] GLU frames before this chunk, replaced
  int B, Tq, NS;
};

constexpr int SC_A_BYTES = BM * D * 4;            // attention-output tile, then fp32 tiles
constexpr int SC_T_OFF = SC_A_BYTES;              // LayerNorm tile, then the GLU tile
constexpr int SC_DWT_OFF = SC_T_OFF + BM * D * 2; // depthwise output tile
constexpr int SC_DW_OFF = SC_DWT_OFF + BM * D * 2;
constexpr int SC_PRM_OFF = SC_DW_OFF + DW_BYTES;  // parameter vectors, CP_* layout
constexpr int SC_BYTES = SC_PRM_OFF + CP_FLOATS * 4;

__global__ __launch_bounds__(NTHREADS) void stream_convmod_kernel(const StreamConvModDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* t_tile = smem + SC_T_OFF;
  char* d_tile = smem + SC_DWT_OFF;
  float* dwl = reinterpret_cast<float*>(smem + SC_DW_OFF);
  const float* prm = reinterpret_cast<const float*>(smem + SC_PRM_OFF);
  auto cidx = [](int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int b0 = blockIdx.x * p.NS;                              // first stream of this workgroup
  const int ns = p.B - b0 < p.NS ? p.B - b0 : p.NS;              // streams it holds
  const int m0 = b0 * p.Tq, nrows = ns * p.Tq;                   // its rows of the token matrix

  // ---- weight stream: 4 steps of Wo, 8 of W1, 4 of W2, four 1 KiB fragments each ----
  constexpr int NS_STEPS = 4 + NSTEP;
  const u32x4* wos = p.Wo + (size_t)wave * 1024 + lane;
  const u32x4* w1s = p.W1 + (size_t)wave * 2048 + lane;
  const u32x4* w2s = p.W2 + (size_t)wave * 1024 + lane;
  u32x4 wr[DEPTH][4];
  auto fetch = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
    if constexpr (s < NS_STEPS) {
      const u32x4* src = s < 4 ? wos + s * 256 : s < 12 ? w1s + (s - 4) * 256 : w2s + (s - 12) * 256;
#pragma unroll
      for (int f = 0; f < 4; ++f) wr[s % DEPTH][f] = src[64 * f];
    }
  };
  // Loads return in issue order and the CU's load path is one queue: what the first phase needs (attention-output
  // tile, depthwise taps, parameter vectors) is requested ahead of the first weights, and every bias or LayerNorm
  // weight is read from LDS later instead of queueing behind the weights in flight.
  u32x4 ov[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + NTHREADS * i;
    ov[i] = u32x4{0u, 0u, 0u, 0u};
    if ((c >> 5) < nrows) ov[i] = *reinterpret_cast<const u32x4*>(p.O + ((size_t)m0 + (c >> 5)) * D + 8 * (c & 31));
  }
  const f32x4 taps_a = *reinterpret_cast<const f32x4*>(p.dw_w9 + 4 * tid);
  f32x4 taps_b = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 64) taps_b = *reinterpret_cast<const f32x4*>(p.dw_w9 + 8 * D + 4 * tid);
  else if (tid < 128) taps_b = *reinterpret_cast<const f32x4*>(p.dw_b + 4 * (tid - 64));
  f32x4 prm_v = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int f = 4 * tid;                      // segments are multiples of 256 floats: uniform per wave
    const float* src = f < CP_LN_B ? p.ln_g + f : f < CP_BO ? p.ln_b + (f - CP_LN_B) : f < CP_B2 ? p.bo + (f - CP_BO)
                       : f < CP_B1 ? p.b2 + (f - CP_B2) : f < CP_FLOATS ? p.b1 + (f - CP_B1) : nullptr;
    if (src) prm_v = *reinterpret_cast<const f32x4*>(src);
  }
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
  __builtin_amdgcn_sched_barrier(0);
  *reinterpret_cast<f32x4*>(dwl + 4 * tid) = taps_a;
  if (tid < 128) *reinterpret_cast<f32x4*>(dwl + 8 * D + 4 * tid) = taps_b;
  if (4 * tid < CP_FLOATS) *reinterpret_cast<f32x4*>(smem + SC_PRM_OFF + 16 * tid) = prm_v;

  // ---- attention out_proj (model.py:429): O tile -> LDS, Wo against it, fp32 result through LDS ----
  f32x4 xin[2][4];
  {
    char* o_tile = smem;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NTHREADS * i;
      *reinterpret_cast<u32x4*>(o_tile + wide_off(c >> 5, c & 31)) = ov[i];
    }
    __syncthreads();
    // the fp32 rows of the stream, wanted when the projection is done
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = wave * 8 + 4 * ps + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xin[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < nrows) xin[ps][k] = *reinterpret_cast<const f32x4*>(p.X + ((size_t)m0 + row) * D + 4 * r + 64 * k);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc0[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 of[2][4];
    auto frames0 = [&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          of[ks & 1][j] = *reinterpret_cast<const bf16x8*>(o_tile + wide_off(16 * j + r, 4 * ks + g));
      }
    };
    frames0(std::integral_constant<int, 0>{});
    static_for<0, 8>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int s = ks >> 1;
      __builtin_amdgcn_sched_barrier(0);
      frames0(std::integral_constant<int, ks + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          mma(acc0[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][2 * (ks & 1) + i]), of[ks & 1][j]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks & 1) fetch(std::integral_constant<int, s + DEPTH>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();       // the O tile has been read; its space now takes the projection
    float* ct0 = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(ct0 + cidx(16 * j + r, 32 * wave + 16 * i + 4 * g)) = acc0[i][j];
    __syncthreads();
  }

  // ---- x += out_proj (model.py:667), parked where the projection stood until the last phase adds the module's
  // output in place; LayerNorm of the new rows into the tile.  Wave w owns rows 8w..8w+7 in two passes of four:
  // DPP row g takes row 8w + 4*pass + g, lane r features 4r + 64k.
  {
    float* ct0 = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = wave * 8 + 4 * ps + g;
      const bool live = row < nrows;
      f32x4 v[4];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * r + 64 * k;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) {
          v[k] = xin[ps][k] + *reinterpret_cast<const f32x4*>(ct0 + cidx(row, col)) +
                 *reinterpret_cast<const f32x4*>(prm + CP_BO + col);
          *reinterpret_cast<f32x4*>(ct0 + cidx(row, col)) = v[k];
        }
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
      }
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
        const f32x4 gg = *reinterpret_cast<const f32x4*>(prm + CP_LN_G + 4 * r + 64 * k);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(prm + CP_LN_B + 4 * r + 64 * k);
        const f32x4 y = (v[k] - mean) * rstd * gg + bb;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
        *reinterpret_cast<bf16x4*>(t_tile + wide_off(row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
      }
    }
  }
  __syncthreads();

  // ---- GEMM-1: acc1[half c][value / gate][row tile]: channels 128c + 16w + 4g.., rows 16j + r ----
  {
    f32x4 acc1[2][2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_for<0, 8>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 tf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) tf[j] = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(16 * j + r, 4 * s + g));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 2; ++i) mma(acc1[c][i][j], __builtin_bit_cast(bf16x8, wr[(4 + s) % DEPTH][2 * c + i]), tf[j]);
      __builtin_amdgcn_sched_barrier(0);
      fetch(std::integral_constant<int, 4 + s + DEPTH>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();      // nobody reads the LayerNorm tile any more
    // bias + GLU on the accumulators -> tile[row][channel] (model.py:520-521)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = 128 * c + 16 * wave + 4 * g;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(prm + CP_B1 + 256 * c + 32 * wave + 4 * g);
      const f32x4 bg = *reinterpret_cast<const f32x4*>(prm + CP_B1 + 256 * c + 32 * wave + 16 + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = acc1[c][0][j] + bv;
        const f32x4 gt = acc1[c][1][j] + bg;
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)(v[k] * sigmoid_f(gt[k]));
        *reinterpret_cast<bf16x4*>(t_tile + wide_off(16 * j + r, ch >> 3) + 2 * (ch & 7)) = o;
      }
    }
  }
  __syncthreads();

  // ---- causal depthwise conv (BatchNorm folded) + SiLU over [cache | chunk] per stream (model.py:236-249, :526-538):
  // item = one stream x 8 channels; a 9-frame window slides down the stream's Tq rows; what is left in the window
  // at the end (frames Tq .. Tq+7 of the sequence) is the stream's next cache
#pragma unroll 1
  for (int it = tid; it < ns * 32; it += NTHREADS) {
    const int sl = it >> 5, cg = it & 31;
    // the taps are re-read from LDS for every frame: holding them next to the window and the weight ring would spill
    float bias[8], win[KW][8];
    {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(dwl + KW * D + 8 * cg);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(dwl + KW * D + 8 * cg + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bias[q] = w0[q];
        bias[4 + q] = w1[q];
      }
    }
    bf16* crow = p.cache + ((size_t)(b0 + sl) * 8) * D + 8 * cg;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(crow + (size_t)m * D);
#pragma unroll
      for (int q = 0; q < 8; ++q) win[m][q] = (float)v[q];
    }
#pragma unroll 1
    for (int i = 0; i < p.Tq; ++i) {
      const int row = sl * p.Tq + i;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(t_tile + wide_off(row, cg));
      float y[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        win[8][q] = (float)v[q];
        y[q] = bias[q];
      }
#pragma unroll
      for (int m = 0; m < KW; ++m) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(dwl + m * D + 8 * cg + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          y[q] = fmaf(w0[q], win[m][q], y[q]);
          y[4 + q] = fmaf(w1[q], win[m][4 + q], y[4 + q]);
        }
      }
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (bf16)silu_f(y[q]);
      *reinterpret_cast<bf16x8*>(d_tile + wide_off(row, cg)) = o;
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int q = 0; q < 8; ++q) win[m][q] = win[m + 1][q];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (bf16)win[m][q];
      *reinterpret_cast<bf16x8*>(crow + (size_t)m * D) = o;
    }
  }
  // rows past the workgroup's last stream hold no data: keep them finite for the MFMAs below
  for (int c = tid; c < (BM - nrows) * 32; c += NTHREADS)
    *reinterpret_cast<u32x4*>(d_tile + wide_off(nrows + (c >> 5), c & 31)) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- GEMM-2: W2 (already in registers or in flight) against the depthwise tile, then the residual add ----
  int lane_b = lane;
  asm volatile("" : "+v"(lane_b));
  const int g2 = lane_b >> 4, r2 = lane_b & 15;
  f32x4 acc2[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 hf[2][4];
  auto frames2 = [&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    if constexpr (ks < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hf[ks & 1][j] = *reinterpret_cast<const bf16x8*>(d_tile + wide_off(16 * j + r2, 4 * ks + g2));
    }
  };
  frames2(std::integral_constant<int, 0>{});
  static_for<0, 8>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    constexpr int s = 12 + (ks >> 1);
    __builtin_amdgcn_sched_barrier(0);
    frames2(std::integral_constant<int, ks + 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mma(acc2[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][2 * (ks & 1) + i]), hf[ks & 1][j]);
    __builtin_amdgcn_sched_barrier(0);
  });
  float* ct = reinterpret_cast<float*>(smem);       // the fp32 region: holds x + out_proj of the live rows
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * j + r2, 32 * wave + 16 * i + 4 * g2)) +=
          acc2[i][j] + *reinterpret_cast<const f32x4*>(prm + CP_B2 + 32 * wave + 16 * i + 4 * g2);
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g2;
    if (row >= nrows) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r2 + 64 * k;
      *reinterpret_cast<f32x4*>(p.X + ((size_t)m0 + row) * D + col) = *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
    }
  }
}

// Row-major W1 [512][256] (GLU-interleaved rows) and W2 [256][256] -> streaming order.  With lane
// = 16g + r holding 8 consecutive K elements,
//   W1p vector ((((w*8 + ks)*2 + c)*2 + i)*64 + lane): row 256c + 32w + 16i + r, columns 32ks + 8g ..
//   W2p vector ((( w*8 + ks)        *2 + i)*64 + lane): row        32w + 16i + r, columns 32ks + 8g ..
__global__ __launch_bounds__(256) void convmod_pack_kernel(const u32x4* __restrict__ W1, const u32x4* __restrict__ W2,
                                                           u32x4* __restrict__ W1p, u32x4* __restrict__ W2p) {
  const int v = blockIdx.x * 256 + threadIdx.x;      // 0 .. 512*256/8 - 1
  const int lane = v & 63, g = lane >> 4, r = lane & 15;
  {
    const int i = (v >> 6) & 1, c = (v >> 7) & 1, ks = (v >> 8) & 7, w = v >> 11;
    W1p[v] = W1[(size_t)(256 * c + 32 * w + 16 * i + r) * (D / 8) + 4 * ks + g];
  }
  if (v < D * D / 8) {
    const int i = (v >> 6) & 1, ks = (v >> 7) & 7, w = v >> 10;
    W2p[v] = W2[(size_t)(32 * w + 16 * i + r) * (D / 8) + 4 * ks + g];
  }
}

}  // namespace

static bool narrow(int dtype, int d, int kernel_size) { return dtype == DT_BF16 && d == D && kernel_size == KW; }

bool convmod_fused_supported(int dtype, int d, int kernel_size) {
  return narrow(dtype, d, kernel_size) || convmod_wide_supported(dtype, d, kernel_size);
}

// the d = 256 kernel can also carry the attention's out_proj (and, behind it, the feed-forward modules)
bool convmod_oproj_supported(int dtype, int d, int kernel_size, int heads) {
  return narrow(dtype, d, kernel_size) && heads * 64 == D;
}


bool stream_convmod_supported(int dtype, int d, int heads, int Tq) {
  return narrow(dtype, d, KW) && heads * 64 == D && Tq >= 1 && Tq <= 16;
}

int launch_stream_convmod(const StreamConvModArgs& a, hipStream_t stream) {
  if (!stream_convmod_supported(DT_BF16, a.d, a.heads, a.Tq)) return fail("stream_convmod: unsupported shape d=%d heads=%d Tq=%d", a.d, a.heads, a.Tq);
  if (a.B <= 0) return fail("stream_convmod: empty batch");
  if (!a.X || !a.O || !a.Wo || !a.bo || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.dw_w9 || !a.dw_b || !a.W2 || !a.b2 || !a.cache)
    return fail("stream_convmod: null operand");
  StreamConvModDev d{};
  d.X = a.X; d.O = (const bf16*)a.O; d.Wo = (const u32x4*)a.Wo; d.bo = a.bo; d.ln_g = a.ln_g; d.ln_b = a.ln_b;
  d.W1 = (const u32x4*)a.W1; d.b1 = a.b1; d.dw_w9 = a.dw_w9; d.dw_b = a.dw_b; d.W2 = (const u32x4*)a.W2; d.b2 = a.b2;
  d.cache = (bf16*)a.cache; d.B = a.B; d.Tq = a.Tq; d.NS = BM / a.Tq;
  if (reserve_lds(reinterpret_cast<const void*>(stream_convmod_kernel), SC_BYTES) != hipSuccess)
    return fail("stream_convmod: cannot reserve %d bytes of LDS", SC_BYTES);
  hipLaunchKernelGGL(stream_convmod_kernel, dim3((a.B + d.NS - 1) / d.NS), dim3(NTHREADS), SC_BYTES, stream, d);
  return check_launch("stream_convmod");
}

int launch_convmod_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, hipStream_t stream) {
  if (!convmod_fused_supported(DT_BF16, d, KW)) return fail("convmod_pack: unsupported width d=%d", d);
  if (!W1 || !W2 || !W1p || !W2p) return fail("convmod_pack: null operand");
  if (W1 == W1p || W2 == W2p) return fail("convmod_pack: packing cannot be done in place");
  if (d != D) return launch_convmod_wide_pack(W1, W2, W1p, W2p, d, stream);
  hipLaunchKernelGGL(convmod_pack_kernel, dim3(2 * D * D / 8 / 256), dim3(256), 0, stream, (const u32x4*)W1,
                     (const u32x4*)W2, (u32x4*)W1p, (u32x4*)W2p);
  return check_launch("convmod_pack");
}

int launch_convmod_fused(const ConvModArgs& a, hipStream_t stream) {
  if (!convmod_fused_supported(DT_BF16, a.d, KW)) return fail("convmod_fused: unsupported width d=%d", a.d);
  if (a.M <= 0 || a.Tp <= 0 || a.Tp % BM || a.M % a.Tp) return fail("convmod_fused: M=%d Tp=%d must be multiples of %d frames", a.M, a.Tp, BM);
  if (a.M > (1 << 20)) return fail("convmod_fused: M=%d exceeds 2^20 frames (32-bit buffer offsets)", a.M);
  if (a.X == a.Xout) return fail("convmod_fused: the output stream must not alias the input stream");
  if (!a.X || !a.Xout || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.dw_w9 || !a.dw_b || !a.W2 || !a.b2) return fail("convmod_fused: null operand");
  if (a.d != D) return launch_convmod_wide(a, stream);
  ConvModDev d{};
  d.X = a.X; d.Xout = a.Xout; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const u32x4*)a.W1; d.b1 = a.b1; d.dw_w9 = a.dw_w9; d.dw_b = a.dw_b;
  d.W2 = (const u32x4*)a.W2; d.b2 = a.b2; d.M = a.M; d.Tp = a.Tp;
  const dim3 grid(a.M / BM), block(NTHREADS);
  auto launch = [&](auto kernel, int lds) {
#ifdef FCENC_STAMPS
    lds = STAMP_OFF + STAMP_BYTES;
#endif
    if (reserve_lds(reinterpret_cast<const void*>(kernel), lds) !=
        hipSuccess)
      return fail("convmod_fused: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, d);
    return 0;
  };
  if (a.O || a.Wo || a.bo) {
    if (!a.O || !a.Wo || !a.bo) return fail("convmod_fused: the fused out_proj needs O, Wo and bo");
    d.O = (const bf16*)a.O; d.Wo = (const u32x4*)a.Wo; d.bo = a.bo;
  }
  if (a.ffn) {
    // the block's second feed-forward module and closing LayerNorm on the module's output, same launch
    const FfnArgs& f = *a.ffn;
    if (!a.O) return fail("convmod_fused: the feed-forward tail is built together with the fused out_proj");
    if (!ffn_fused_supported(DT_BF16, f.d, f.ff) || f.M != a.M) return fail("convmod_fused: unsupported feed-forward tail");
    if (!f.ln_g || !f.ln_b || !f.W1 || !f.b1 || !f.W2 || !f.b2 || f.Wq || f.Xc) return fail("convmod_fused: bad feed-forward tail operands");
    if ((f.post_g == nullptr) != (f.post_b == nullptr)) return fail("convmod_fused: closing LayerNorm affine incomplete");
    if (f.compact && (f.Tp <= 0 || f.Tvalid <= 0 || f.Tvalid > f.Tp || f.M % f.Tp)) return fail("convmod_fused: compact geometry");
    ffnk::FfnDev& q = d.ffn;
    q.X = a.Xout; q.ln_g = f.ln_g; q.ln_b = f.ln_b; q.W1 = (const u32x4*)f.W1; q.b1 = f.b1; q.W2 = (const u32x4*)f.W2;
    q.b2 = f.b2; q.post_g = f.post_g; q.post_b = f.post_b; q.Xc = nullptr; q.compact = f.compact; q.Wq = nullptr;
    q.bq = nullptr; q.Q = nullptr; q.M = f.M; q.Tp = f.Tp > 0 ? f.Tp : 1; q.Tvalid = f.Tvalid; q.alpha = f.alpha;
    if (a.next) {
      // ... and the next block's first feed-forward module + q/k/v projection
      const FfnArgs& n = *a.next;
      if (n.ff != f.ff || n.d != f.d || n.M != a.M || !n.ln_g || !n.ln_b || !n.W1 || !n.b1 || !n.W2 || !n.b2 || !n.Wq ||
          !n.bq || !n.Q || n.nq != 768 || n.post_g || n.post_b || n.compact || n.Xc || f.compact)
        return fail("convmod_fused: bad operands for the chained feed-forward module");
      ffnk::FfnDev& z = d.next;
      z.X = a.Xout; z.ln_g = n.ln_g; z.ln_b = n.ln_b; z.W1 = (const u32x4*)n.W1; z.b1 = n.b1; z.W2 = (const u32x4*)n.W2;
      z.b2 = n.b2; z.post_g = nullptr; z.post_b = nullptr; z.Xc = nullptr; z.compact = nullptr; z.Wq = (const u32x4*)n.Wq;
      z.bq = n.bq; z.Q = (bf16*)n.Q; z.M = n.M; z.Tp = 1; z.Tvalid = 0; z.alpha = n.alpha;
      if (f.ff == 1024) { if (launch(convmod_fused_kernel<true, 4, true>, Lds<true>::bytes(4, true))) return -1; }
      else { if (launch(convmod_fused_kernel<true, 8, true>, Lds<true>::bytes(8, true))) return -1; }
    } else if (f.ff == 1024) { if (launch(convmod_fused_kernel<true, 4>, Lds<true>::bytes(4, false))) return -1; }
    else { if (launch(convmod_fused_kernel<true, 8>, Lds<true>::bytes(8, false))) return -1; }
  } else if (a.O) {
    if (launch(convmod_fused_kernel<true, 0>, Lds<true>::BYTES)) return -1;
  } else {
    if (launch(convmod_fused_kernel<false, 0>, Lds<false>::BYTES)) return -1;
  }
  return check_launch("convmod_fused");
}

#ifdef FCENC_STAMPS
// diagnostic builds only: the phase clocks of the last chained block launch, [workgroup][wave][64 slots]
extern "C" int fcenc_debug_stamps(unsigned long long* host, int nwg) {
  if (nwg > STAMP_MAX_WG) nwg = STAMP_MAX_WG;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)nwg * 512 * 8) == hipSuccess ? nwg : -1;
}
#endif

}  // namespace fcenc
