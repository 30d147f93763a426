// Fused convolution module of a conformer block for the wider models (bf16 MFMA, d_model = 384 or 512, kernel 9;
// the d = 256 kernel, which also carries out_proj and the feed-forward modules, lives in convmod.hip):
//
//     xout = x + W2 silu(dw9(glu(W1 LN(x) + b1)) + bdw) + b2        (model.py:516-544, :670-671)
//
// in ONE kernel; dw9 is the causal depthwise convolution with the eval-mode BatchNorm folded in (model.py:236-249).
// A workgroup (8 waves) owns 64 frames of one utterance plus the 8 frames in front of them that the depthwise taps
// reach back to (recomputed, not exchanged):
//
//   1. LayerNorm of the 64 + 8 frames into an 80-row bf16 LDS tile T (rows 0-7 unused, 8-15 halo).
//   2. GEMM-1 in D/128 passes of 256 GLU-interleaved rows of W1 (value | gate in 16-row groups): 10 accumulators
//      per wave and pass, bias + GLU on them, 128 channels of the result into a second 80-row tile G.
//   3. Depthwise conv + SiLU: a thread slides a 12-row window of G over 8 channels of 4 frames and writes the 64
//      result rows into T (free since GEMM-1).
//   4. GEMM-2: W2 against the 64 rows of T, then the residual add through an fp32 LDS tile.
//
// Weights never touch LDS: each wave streams the rows it multiplies (32 of every 256-row block of W1, D/8 rows of
// W2) from L2 straight into MFMA A-fragments through a ring of 16 fragment registers, stored in consumption order
// by convmod_wide_pack_kernel.  The depthwise taps are read from global memory (L2 hits): at d = 512 the two
// tiles fill the CU's 160 KiB of LDS.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int BM = 64;            // frames per workgroup
constexpr int HALO = 16;          // rows in front of them in the tiles (only the last 8 are used)
constexpr int ROWS = BM + HALO;
constexpr int KW = 9;             // depthwise taps
constexpr int NTHREADS = 512;
constexpr int NF = 16;            // weight fragments in flight per wave

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int D>
struct Geo {
  static constexpr int KS = D / 32;         // k32 steps of either GEMM (K = D)
  static constexpr int NB = D / 128;        // 256-row blocks of W1 (2D rows), 128 GLU channels each
  static constexpr int RW = D / 128;        // 16-row fragments of W2 per wave (D/8 output rows)
  static constexpr int F1 = 2 * KS * NB;    // fragments per wave, GEMM-1
  static constexpr int F2 = KS * RW;        //                     GEMM-2
  static constexpr int ROWB = D * 2;
  static constexpr int T_BYTES = ROWS * ROWB;
  static constexpr int C_BYTES = BM * D * 4;
  static constexpr int LDS = 2 * T_BYTES > C_BYTES ? 2 * T_BYTES : C_BYTES;
  static constexpr int NK = D / 64;
};

// Lane coordinates behind an optimisation barrier, taken afresh at the start of every phase: otherwise the compiler
// keeps the LDS / global addresses it formed from them in one phase alive, and spilled, across the whole kernel.
struct LaneId {
  int g, r;
  __device__ __forceinline__ LaneId() {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    g = lane >> 4;
    r = lane & 15;
  }
};

template <int ROWB>
__device__ __forceinline__ uint32_t tile_at(int row, int chunk) {
  return (uint32_t)row * ROWB + (uint32_t)((chunk ^ (row & 15)) << 4);
}

struct WideConvDev {
  const float* X;           // [M][D] fp32 residual stream
  float* Xout;              // [M][D] updated stream; a different buffer: a workgroup reads the 8 frames in front of
                            // its own, which its neighbour rewrites
  const float* ln_g;
  const float* ln_b;
  const u32x4* W1;          // packed, see convmod_wide_pack_kernel
  const float* b1;          // [2D], rows interleaved 16 value / 16 gate
  const float* dw_w9;       // [9][D] depthwise taps, BatchNorm folded
  const float* dw_b;        // [D]
  const u32x4* W2;          // packed
  const float* b2;          // [D]
  int M, Tp;
};

template <int D>
__global__ __launch_bounds__(NTHREADS) void convmod_wide_kernel(const WideConvDev p) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* t_tile = smem;
  char* g_tile = smem + G::T_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const bool first = (m0 % p.Tp) == 0;        // utterance starts here: the halo is the causal zero padding

  // ---- weight stream: positions 0 .. F1-1 are GEMM-1 ((block b, step ks, value / gate i) in that order),
  // F1 .. F1+F2-1 GEMM-2 ((step ks, fragment i)); position P lives in ring slot P % NF
  const u32x4* w1s = p.W1 + (size_t)wave * G::F1 * 64 + lane;
  const u32x4* w2s = p.W2 + (size_t)wave * G::F2 * 64 + lane;
  u32x4 wr[NF];
  auto fetch = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    if constexpr (P < G::F1) wr[P % NF] = w1s[P * 64];
    else if constexpr (P < G::F1 + G::F2) wr[P % NF] = w2s[(P - G::F1) * 64];
  };
  // ---- LayerNorm.  Wave w owns frames 8w..8w+7 in two passes of four: DPP row g (16 lanes) takes frame
  // 8w + 4*pass + g and lane r of the row features 4r + 64k (k < D/64).  Waves 0 and 1 then take the 8 halo frames.
  // Loads return in issue order, so the rows and the affine are requested before the first weights.
  {
  const LaneId id;
  const int g = id.g, r = id.r;
  const bool do_halo = wave < 2 && !first;
  f32x4 xin[2][G::NK], ln_gv[G::NK], ln_bv[G::NK];
  if (do_halo) {
#pragma unroll
    for (int k = 0; k < G::NK; ++k)
      xin[0][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)(m0 - 8 + 4 * wave + g) * D + 4 * r + 64 * k);
  }
#pragma unroll
  for (int k = 0; k < G::NK; ++k) {
    ln_gv[k] = *reinterpret_cast<const f32x4*>(p.ln_g + 4 * r + 64 * k);
    ln_bv[k] = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * r + 64 * k);
  }
  auto layer_norm_row = [&](const f32x4 (&v)[G::NK], int tile_row) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    const float mean = row16_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const f32x4 c = v[k] - mean;
      q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    }
    const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const f32x4 y = (v[k] - mean) * rstd * ln_gv[k] + ln_bv[k];
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
      *reinterpret_cast<bf16x4*>(t_tile + tile_at<G::ROWB>(tile_row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
    }
  };
  // the halo frames first (waves 0 and 1; one exposed round trip), so that their registers take the own frames next
  if (do_halo) layer_norm_row(xin[0], 8 + 4 * wave + g);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
#pragma unroll
    for (int k = 0; k < G::NK; ++k)
      xin[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)(m0 + wave * 8 + 4 * ps + g) * D + 4 * r + 64 * k);
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, NF>([&](auto pc) { fetch(pc); });
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) layer_norm_row(xin[ps], HALO + wave * 8 + 4 * ps + g);
  }
  __syncthreads();

  // ---- GEMM-1, one 256-row block of W1 per pass: acc1[value / gate][frame tile], channels 128b + 16w + 4g..,
  // frames 16j + r - 16; bias + GLU on the accumulators -> G[frame row][channel] (model.py:520-521); the halo
  // rows of an utterance's first workgroup are the causal zero padding of the depthwise conv
  static_for<0, G::NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const LaneId id;
    const int g = id.g, r = id.r;
    f32x4 acc1[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the pass's bias now: requested behind the ring it would otherwise wait for the whole of it
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b1 + 256 * b + 32 * wave + 4 * g);
    const f32x4 bg = *reinterpret_cast<const f32x4*>(p.b1 + 256 * b + 32 * wave + 16 + 4 * g);
    static_for<0, G::KS>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      constexpr int P0 = (b * G::KS + ks) * 2;
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 tf[5];
#pragma unroll
      for (int j = 0; j < 5; ++j)
        tf[j] = *reinterpret_cast<const bf16x8*>(t_tile + tile_at<G::ROWB>(16 * j + r, 4 * ks + g));
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) mma(acc1[i][j], __builtin_bit_cast(bf16x8, wr[(P0 + i) % NF]), tf[j]);
      __builtin_amdgcn_sched_barrier(0);
      fetch(std::integral_constant<int, P0 + NF>{});
      fetch(std::integral_constant<int, P0 + 1 + NF>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    const int ch = 128 * b + 16 * wave + 4 * g;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const f32x4 v = acc1[0][j] + bv;
      const f32x4 gt = acc1[1][j] + bg;
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (bf16)((j == 0 && first) ? 0.f : v[k] * sigmoid_f(gt[k]));
      *reinterpret_cast<bf16x4*>(g_tile + tile_at<G::ROWB>(16 * j + r, ch >> 3) + 2 * (ch & 7)) = o;
    }
  });
  __syncthreads();      // G is complete; nobody reads the LayerNorm tile any more

  // ---- causal depthwise conv (BatchNorm folded) + SiLU (model.py:526-538): item = 8 channels x 4 frames,
  // window rows from G, the 64 result rows into T
  {
    constexpr int NCG = D / 8;                       // channel groups
    constexpr int ITEMS = NCG * (BM / 4);
#pragma unroll 1
    for (int it = tid; it < ITEMS; it += NTHREADS) {
      const int cg = it % NCG, tl = it / NCG;
      float w[KW][8], y[4][8];
#pragma unroll
      for (int m = 0; m <= KW; ++m) {
        const float* src = (m < KW ? p.dw_w9 + (size_t)m * D : p.dw_b) + 8 * cg;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + 4);
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
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(g_tile + tile_at<G::ROWB>(8 + 4 * tl + k, cg));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int m = k - s;                      // frame s takes row k through tap m
          if (m < 0 || m >= KW) continue;
#pragma unroll
          for (int q = 0; q < 8; ++q) y[s][q] = fmaf(w[m][q], (float)v[q], y[s][q]);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (bf16)silu_f(y[s][q]);
        *reinterpret_cast<bf16x8*>(t_tile + tile_at<G::ROWB>(4 * tl + s, cg)) = o;
      }
    }
  }
  __syncthreads();

  // ---- GEMM-2: W2 (its first fragments already in the ring) against rows 0..63 of T ----
  const LaneId id2;
  const int g = id2.g, r = id2.r;
  f32x4 acc2[G::RW][4];      // output features (D/8)w + 16i + 4g.., frames 16j + r
#pragma unroll
  for (int i = 0; i < G::RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 hf[2][4];
  auto frames2 = [&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    if constexpr (ks < G::KS) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hf[ks & 1][j] = *reinterpret_cast<const bf16x8*>(t_tile + tile_at<G::ROWB>(16 * j + r, 4 * ks + g));
    }
  };
  frames2(std::integral_constant<int, 0>{});
  static_for<0, G::KS>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    constexpr int P0 = G::F1 + ks * G::RW;
    __builtin_amdgcn_sched_barrier(0);
    frames2(std::integral_constant<int, ks + 1>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, G::RW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) mma(acc2[i][j], __builtin_bit_cast(bf16x8, wr[(P0 + i) % NF]), hf[ks & 1][j]);
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, G::RW>([&](auto ic) { fetch(std::integral_constant<int, P0 + decltype(ic)::value + NF>{}); });
    __builtin_amdgcn_sched_barrier(0);
  });

  // ---- epilogue: out tile [64 frames][D features] fp32 through LDS (over both tiles), residual add ----
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  auto cidx = [](int row, int col) { return row * D + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };
#pragma unroll
  for (int i = 0; i < G::RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * j + r, (D / 8) * wave + 16 * i + 4 * g)) = acc2[i][j];
  __syncthreads();
  // every global read of the epilogue first (issued before the barrier above would be better still, but the
  // registers are the accumulators'): a load behind a store to Xout would wait for it, one round trip per vector
  const LaneId id3;
  f32x4 xres[2][G::NK], b2v[G::NK];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
#pragma unroll
    for (int k = 0; k < G::NK; ++k)
      xres[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)(m0 + wave * 8 + 4 * ps + id3.g) * D + 4 * id3.r + 64 * k);
#pragma unroll
  for (int k = 0; k < G::NK; ++k) b2v[k] = *reinterpret_cast<const f32x4*>(p.b2 + 4 * id3.r + 64 * k);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int g = id3.g, r = id3.r;
    const int row = wave * 8 + 4 * ps + g;
    const int m = m0 + row;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const int col = 4 * r + 64 * k;
      *reinterpret_cast<f32x4*>(p.Xout + (size_t)m * D + col) =
          xres[ps][k] + *reinterpret_cast<const f32x4*>(ct + cidx(row, col)) + b2v[k];
    }
  }
}

// Row-major W1 [2D][D] (GLU-interleaved rows) and W2 [D][D] -> streaming order.  With lane = 16g + r holding 8
// consecutive K elements,
//   W1p vector ((((w*NB + b)*KS + ks)*2 + i)*64 + lane): row 256b + 32w + 16i + r, columns 32ks + 8g ..
//   W2p vector ((( w*KS + ks)*RW + i)*64 + lane):        row (D/8)w + 16i + r,     columns 32ks + 8g ..
template <int D>
__global__ __launch_bounds__(256) void convmod_wide_pack_kernel(const u32x4* __restrict__ W1, const u32x4* __restrict__ W2,
                                                                u32x4* __restrict__ W1p, u32x4* __restrict__ W2p) {
  using G = Geo<D>;
  const int v = blockIdx.x * 256 + threadIdx.x;      // 0 .. 2D*D/8 - 1
  if (v >= 2 * D * D / 8) return;
  const int lane = v & 63, g = lane >> 4, r = lane & 15;
  {
    int u = v >> 6;
    const int i = u % 2; u /= 2;
    const int ks = u % G::KS; u /= G::KS;
    const int b = u % G::NB, w = u / G::NB;
    W1p[v] = W1[(size_t)(256 * b + 32 * w + 16 * i + r) * (D / 8) + 4 * ks + g];
  }
  if (v < D * D / 8) {
    int u = v >> 6;
    const int i = u % G::RW; u /= G::RW;
    const int ks = u % G::KS, w = u / G::KS;
    W2p[v] = W2[(size_t)((D / 8) * w + 16 * i + r) * (D / 8) + 4 * ks + g];
  }
}

template <int D>
int launch_wide(const WideConvDev& d, hipStream_t stream) {
  auto k = convmod_wide_kernel<D>;
  constexpr int lds = Geo<D>::LDS;
  if (reserve_lds(reinterpret_cast<const void*>(k), lds) != hipSuccess)
    return fail("convmod_fused: cannot reserve %d bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3(d.M / BM), dim3(NTHREADS), lds, stream, d);
  return check_launch("convmod_fused (wide)");
}

}  // namespace

bool convmod_wide_supported(int dtype, int d, int kernel_size) {
  return dtype == DT_BF16 && (d == 384 || d == 512) && kernel_size == KW;
}

int launch_convmod_wide_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, hipStream_t stream) {
  const int nvec = 2 * d * d / 8;
  const dim3 grid((nvec + 255) / 256), block(256);
  if (d == 384)
    hipLaunchKernelGGL(convmod_wide_pack_kernel<384>, grid, block, 0, stream, (const u32x4*)W1, (const u32x4*)W2, (u32x4*)W1p, (u32x4*)W2p);
  else
    hipLaunchKernelGGL(convmod_wide_pack_kernel<512>, grid, block, 0, stream, (const u32x4*)W1, (const u32x4*)W2, (u32x4*)W1p, (u32x4*)W2p);
  return check_launch("convmod_pack (wide)");
}

int launch_convmod_wide(const ConvModArgs& a, hipStream_t stream) {
  if (a.O || a.Wo || a.bo || a.ffn || a.next) return fail("convmod_fused: d = %d carries the module alone (no out_proj / feed-forward tail)", a.d);
  WideConvDev d{};
  d.X = a.X; d.Xout = a.Xout; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const u32x4*)a.W1; d.b1 = a.b1; d.dw_w9 = a.dw_w9;
  d.dw_b = a.dw_b; d.W2 = (const u32x4*)a.W2; d.b2 = a.b2; d.M = a.M; d.Tp = a.Tp;
  if (a.d == 384) return launch_wide<384>(d, stream);
  return launch_wide<512>(d, stream);
}

}  // namespace fcenc
