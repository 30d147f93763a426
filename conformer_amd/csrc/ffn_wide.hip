// Fused macaron feed-forward block for the wider models (bf16 MFMA, d_model = 384 or 512; the d = 256 kernel
// with its q/k/v projection lives in ffn.hip):
//
//     x <- x + alpha * (W2 silu(W1 LN(x) + b1) + b2)            (model.py:577-584, :663, :674)
//     optionally followed by the block's closing LayerNorm      (model.py:677)
//
// Same plan as the d = 256 kernel: a workgroup (8 waves) owns 64 tokens, normalises them into an LDS tile and walks
// the hidden dimension in chunks of 256 units; GEMM-1 (D/32 k32 steps of W1) produces a chunk, bias + SiLU run on
// the accumulators, the bf16 chunk goes to a second LDS tile and GEMM-2 (8 k32 steps of W2) accumulates it into the
// D x 64 output that stays in registers for the whole kernel.  The (M, 4d) hidden activation never leaves the CU.
//
// Weights never touch LDS: wave w multiplies hidden rows 32w..32w+31 of a W1 chunk and output rows
// (D/8)w..(D/8)(w+1)-1 of W2, so it streams exactly those rows from L2 into MFMA A-fragments through a ring of 16
// fragment registers (16 KiB per wave, 128 KiB per CU in flight).  The weights are stored in the order each wave
// consumes them (ffn_wide_pack_kernel), so the stream is one linear walk of 1 KiB loads per wave.
//
// What differs from d = 256: GEMM-2 steps carry D/128 (3 or 4) fragments per wave instead of 2, so the ring counts
// fragments, not steps; the fp32 input rows are re-read for the residual instead of being held in registers (48-64
// registers at these widths); the tiles are D*128 + 32 KiB, the epilogue's fp32 tile D*256 bytes (96 / 128 KiB).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int BM = 64;            // tokens per workgroup
constexpr int HC = 256;           // hidden units per chunk
constexpr int NTHREADS = 512;
constexpr int NF = 16;            // weight fragments (1 KiB per wave each) in flight per wave

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int D>
struct Geo {
  static constexpr int KS1 = D / 32;        // k32 steps of GEMM-1 (K = D)
  static constexpr int RW = D / 128;        // 16-row fragments of GEMM-2 per wave (D/8 output rows)
  static constexpr int F1 = 2 * KS1;        // fragments per wave per chunk, GEMM-1
  static constexpr int F2 = 8 * RW;         //                               GEMM-2
  static constexpr int F = F1 + F2;
  static constexpr int ROWB = D * 2;        // bytes per row of the normalised-input tile
  static constexpr int A_BYTES = BM * ROWB;
  static constexpr int H_BYTES = BM * HC * 2;
  static constexpr int C_BYTES = BM * D * 4;
  static constexpr int LDS = (A_BYTES + 2 * H_BYTES) > C_BYTES ? (A_BYTES + 2 * H_BYTES) : C_BYTES;   // two hidden tiles
  static constexpr int NK = D / 64;         // float4 per lane per token row (16 lanes x 4 features x NK)
  static_assert(F % NF == 0 || true, "");
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

// [64 tokens][W bf16] tiles with ROWB-byte rows (a multiple of 256): 16-byte chunk c of row m at c ^ (m & 15);
// every row starts on bank 0, the XOR spreads the 16 rows of a fragment read over the 16 slots of a bank row
template <int ROWB>
__device__ __forceinline__ uint32_t tile_at(int row, int chunk) {
  return (uint32_t)row * ROWB + (uint32_t)((chunk ^ (row & 15)) << 4);
}

struct WideDev {
  float* X;                 // [M][D] fp32 residual stream, updated in place
  const float* ln_g;
  const float* ln_b;
  const u32x4* W1;          // packed, see ffn_wide_pack_kernel
  const float* b1;          // [FF]
  const u32x4* W2;          // packed
  const float* b2;          // [D]
  const float* post_g;      // closing LayerNorm affine or null
  const float* post_b;
  bf16* Xc;                 // optional bf16 copy of the new X
  float* compact;           // optional [B][Tvalid][D] copy of the new X without padding frames
  int M, Tp, Tvalid;
  float alpha;
};

template <int D, int NCHUNK>
__global__ __launch_bounds__(NTHREADS) void ffn_wide_kernel(const WideDev p) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_tile = smem;
  // Hidden chunks alternate between two tiles: a wave may write chunk c (bias + SiLU) while slower waves still read
  // chunk c - 1 in their second product, so one barrier per chunk (the one that publishes it) is enough; chunk c
  // overwrites the tile of chunk c - 2, which every wave had finished reading before it passed the barrier of c - 1.
  // The odd tile's offset goes through an empty asm statement so that it is added where it is used (kept as a second
  // set of address registers it would be spilled).
  auto h_tile_of = [&](int c) {
    uint32_t off = G::A_BYTES + (c & 1) * G::H_BYTES;
    if ((c & 1) || G::A_BYTES >= 65536) asm volatile("" : "+s"(off));
    return smem + off;
  };
  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- weight stream: sequence position P = chunk * F + q; q < F1: GEMM-1 step q / 2, fragment q % 2;
  // q >= F1: GEMM-2 step (q - F1) / RW, fragment (q - F1) % RW.  Position P lives in ring slot P % NF.
  const u32x4* w1s = p.W1 + (size_t)wave * G::F1 * 64 + lane;
  const u32x4* w2s = p.W2 + (size_t)wave * G::F2 * 64 + lane;
  u32x4 wr[NF];
  auto fetch = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    constexpr int c = P / G::F, q = P % G::F;
    if constexpr (c < NCHUNK) {
      if constexpr (q < G::F1) wr[P % NF] = w1s[(size_t)c * 8 * G::F1 * 64 + q * 64];
      else wr[P % NF] = w2s[(size_t)c * 8 * G::F2 * 64 + (q - G::F1) * 64];
    }
  };
  // ---- pre-LayerNorm of the 64 tokens into a_tile.  Wave w owns tokens 8w..8w+7 in two passes of four: DPP row g
  // (16 lanes) takes token 8w + 4*pass + g, lane r of the row features 4r + 64k (k < D/64); row sums are DPP adds.
  // Loads return in issue order, so what the LayerNorm needs (rows, affine) is requested before the first weights.
  f32x4 xin[2][G::NK], ln_gv[G::NK], ln_bv[G::NK];
  {
    const LaneId id;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + id.g;
#pragma unroll
      for (int k = 0; k < G::NK; ++k) {
        xin[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M) xin[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * id.r + 64 * k);
      }
    }
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      ln_gv[k] = *reinterpret_cast<const f32x4*>(p.ln_g + 4 * id.r + 64 * k);
      ln_bv[k] = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * id.r + 64 * k);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, NF>([&](auto pc) { fetch(pc); });
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const LaneId id;
    const int g = id.g, r = id.r;
    const int row = wave * 8 + 4 * ps + g;
    f32x4 (&x)[G::NK] = xin[ps];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) s += x[k][0] + x[k][1] + x[k][2] + x[k][3];
    const float mean = row16_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const f32x4 c = x[k] - mean;
      q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    }
    const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const f32x4 y = (x[k] - mean) * rstd * ln_gv[k] + ln_bv[k];
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
      *reinterpret_cast<bf16x4*>(a_tile + tile_at<G::ROWB>(row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
    }
  }
  __syncthreads();

  const LaneId idm;
  const int g = idm.g, r = idm.r;
  f32x4 acc2[G::RW][4];      // output: features (D/8)w + 16i + 4g.., tokens 16j + r
#pragma unroll
  for (int i = 0; i < G::RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[2][4];          // hidden chunk: units 32w + 16i + 4g.., tokens 16j + r
  f32x4 bias1[2];
  bf16x8 tf[2][4];           // token-side fragments, double-buffered by step parity

  // steps of a chunk: t < KS1 -> GEMM-1 k32 step t on a_tile; t >= KS1 -> GEMM-2 k32 step t - KS1 on h_tile
  constexpr int NT = G::KS1 + 8;
  auto tokens = [&](auto tc, auto bufc, auto chunkc) {
    constexpr int t = decltype(tc)::value, buf = decltype(bufc)::value;
    const char* h_tile = h_tile_of(decltype(chunkc)::value);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (t < G::KS1)
        tf[buf][j] = *reinterpret_cast<const bf16x8*>(a_tile + tile_at<G::ROWB>(16 * j + r, 4 * t + g));
      else
        tf[buf][j] = *reinterpret_cast<const bf16x8*>(h_tile + tile_at<512>(16 * j + r, 4 * (t - G::KS1) + g));
    }
  };

  tokens(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  static_for<0, NCHUNK>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int S = c * NT + t;                    // global step index: token-fragment buffer parity
      constexpr int P0 = c * G::F + (t < G::KS1 ? 2 * t : G::F1 + (t - G::KS1) * G::RW);   // first fragment of the step
      constexpr int NFR = t < G::KS1 ? 2 : G::RW;      // fragments the step consumes
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bias1[i] = *reinterpret_cast<const f32x4*>(p.b1 + c * HC + 32 * wave + 16 * i + 4 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // next step's token fragments (not across the SiLU hand-over, whose tile is not written yet, and not past
      // the end of a chunk's GEMM-2 into the next chunk's GEMM-1 ... that one IS legal: a_tile never changes)
      if constexpr (t != G::KS1 - 1 && !(c == NCHUNK - 1 && t == NT - 1)) {
        constexpr int tn = (t + 1) % NT;
        tokens(std::integral_constant<int, tn>{}, std::integral_constant<int, (S + 1) & 1>{}, cc);   // (a hidden-tile step stays in chunk c)
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NFR>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (t < G::KS1) mma(acc1[i][j], __builtin_bit_cast(bf16x8, wr[(P0 + i) % NF]), tf[S & 1][j]);
          else mma(acc2[i][j], __builtin_bit_cast(bf16x8, wr[(P0 + i) % NF]), tf[S & 1][j]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NFR>([&](auto ic) { fetch(std::integral_constant<int, P0 + decltype(ic)::value + NF>{}); });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t == G::KS1 - 1) {
        // bias + SiLU on the accumulators, bf16 chunk to its hidden tile [token][unit]; the barrier publishes the chunk
        char* const h_tile = h_tile_of(c);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int u = 32 * wave + 16 * i + 4 * g;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc1[i][j] * kLog2e + bias1[i] * kLog2e;   // u = x log2e: one FMA (the scaled bias is loop-invariant)
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16)silu_log2e(v[k]);
            *reinterpret_cast<bf16x4*>(h_tile + tile_at<512>(16 * j + r, u >> 3) + 2 * (u & 7)) = o;
          }
        }
        __syncthreads();
        tokens(std::integral_constant<int, G::KS1>{}, std::integral_constant<int, (S + 1) & 1>{}, cc);
      }
    });
  });

  // ---- epilogue: out tile [64 tokens][D features] fp32 through LDS (reuses both tiles) ----
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  auto cidx = [](int row, int col) { return row * D + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };
#pragma unroll
  for (int i = 0; i < G::RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * j + r, (D / 8) * wave + 16 * i + 4 * g)) = acc2[i][j];
  __syncthreads();

  // same token / feature ownership as the prologue: 256 contiguous bytes per 16-lane row and access.  Everything the
  // two passes read from global memory is requested up front: a load behind the first pass's stores would have to
  // wait for them (the compiler cannot tell that X's rows do not overlap), one round trip per vector.
  f32x4 xres[2][G::NK], b2v[G::NK], post_gv[G::NK], post_bv[G::NK];
  {
    const LaneId ide;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + ide.g;
#pragma unroll
      for (int k = 0; k < G::NK; ++k) {
        xres[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M) xres[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * ide.r + 64 * k);
      }
    }
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      b2v[k] = *reinterpret_cast<const f32x4*>(p.b2 + 4 * ide.r + 64 * k);
      if (p.post_g) {
        post_gv[k] = *reinterpret_cast<const f32x4*>(p.post_g + 4 * ide.r + 64 * k);
        post_bv[k] = *reinterpret_cast<const f32x4*>(p.post_b + 4 * ide.r + 64 * k);
      }
    }
  }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const LaneId ide;
    const int g = ide.g, r = ide.r;
    const int row = wave * 8 + 4 * ps + g;
    const int m = m0 + row;
    if (m >= p.M) continue;             // whole DPP rows drop out together; the row sums below stay inside a row
    f32x4 y[G::NK];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const int col = 4 * r + 64 * k;
      y[k] = xres[ps][k] + (*reinterpret_cast<const f32x4*>(ct + cidx(row, col)) + b2v[k]) * p.alpha;
      s += y[k][0] + y[k][1] + y[k][2] + y[k][3];
    }
    if (p.post_g) {
      const float mean = row16_sum(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < G::NK; ++k) {
        y[k] -= mean;
        q += y[k][0] * y[k][0] + y[k][1] * y[k][1] + y[k][2] * y[k][2] + y[k][3] * y[k][3];
      }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
      for (int k = 0; k < G::NK; ++k)
        y[k] = y[k] * rstd * post_gv[k] + post_bv[k];
    }
    const int bt = m / p.Tp, tt = m - bt * p.Tp;
#pragma unroll
    for (int k = 0; k < G::NK; ++k) {
      const int col = 4 * r + 64 * k;
      *reinterpret_cast<f32x4*>(p.X + (size_t)m * D + col) = y[k];
      if (p.Xc) store4(p.Xc + (size_t)m * D + col, y[k]);
      if (p.compact && tt < p.Tvalid)
        *reinterpret_cast<f32x4*>(p.compact + ((size_t)bt * p.Tvalid + tt) * D + col) = y[k];
    }
  }
}

// Row-major W1 [FF][D] and W2 [D][FF] -> the order the waves consume them.  One 16-byte vector = 8 consecutive K
// elements of one weight row for one lane (16g + r):
//   W1p vector ((((c*8 + w)*KS1 + ks)*2 + i)*64 + lane):  row 256c + 32w + 16i + r of W1, columns 32ks + 8g ..
//   W2p vector ((((c*8 + w)*8 + ks)*RW + i)*64 + lane):   row (D/8)w + 16i + r of W2, columns 256c + 32ks + 8g ..
template <int D>
__global__ __launch_bounds__(256) void ffn_wide_pack_kernel(const u32x4* __restrict__ W1, const u32x4* __restrict__ W2,
                                                            u32x4* __restrict__ W1p, u32x4* __restrict__ W2p, int ff) {
  using G = Geo<D>;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= ff * D / 8) return;
  const int lane = v & 63, g = lane >> 4, r = lane & 15;
  {
    int u = v >> 6;
    const int i = u % 2; u /= 2;
    const int ks = u % G::KS1; u /= G::KS1;
    const int w = u % 8, c = u / 8;
    W1p[v] = W1[(size_t)(256 * c + 32 * w + 16 * i + r) * (D / 8) + 4 * ks + g];
  }
  {
    int u = v >> 6;
    const int i = u % G::RW; u /= G::RW;
    const int ks = u % 8; u /= 8;
    const int w = u % 8, c = u / 8;
    W2p[v] = W2[(size_t)((D / 8) * w + 16 * i + r) * (ff / 8) + 32 * c + 4 * ks + g];
  }
}

template <int D, int NCHUNK>
int launch_wide(const WideDev& d, hipStream_t stream) {
  auto k = ffn_wide_kernel<D, NCHUNK>;
  constexpr int lds = Geo<D>::LDS;
  if (reserve_lds(reinterpret_cast<const void*>(k), lds) != hipSuccess)
    return fail("ffn_fused: cannot reserve %d bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3((d.M + BM - 1) / BM), dim3(NTHREADS), lds, stream, d);
  return check_launch("ffn_fused (wide)");
}

}  // namespace

bool ffn_wide_supported(int dtype, int d, int ff) {
  return dtype == DT_BF16 && ((d == 384 && ff == 1536) || (d == 512 && ff == 2048));
}

int launch_ffn_wide_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, hipStream_t stream) {
  const int nvec = ff * d / 8;
  const dim3 grid((nvec + 255) / 256), block(256);
  if (d == 384)
    hipLaunchKernelGGL(ffn_wide_pack_kernel<384>, grid, block, 0, stream, (const u32x4*)W1, (const u32x4*)W2, (u32x4*)W1p, (u32x4*)W2p, ff);
  else
    hipLaunchKernelGGL(ffn_wide_pack_kernel<512>, grid, block, 0, stream, (const u32x4*)W1, (const u32x4*)W2, (u32x4*)W1p, (u32x4*)W2p, ff);
  return check_launch("ffn_pack (wide)");
}

int launch_ffn_wide(const FfnArgs& a, hipStream_t stream) {
  if (a.Wq || a.partial) return fail("ffn_fused: d = %d has no fused projection / split schedule", a.d);
  WideDev d{};
  d.X = a.X; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const u32x4*)a.W1; d.b1 = a.b1; d.W2 = (const u32x4*)a.W2;
  d.b2 = a.b2; d.post_g = a.post_g; d.post_b = a.post_b; d.Xc = (bf16*)a.Xc; d.compact = a.compact;
  d.M = a.M; d.Tp = a.Tp > 0 ? a.Tp : 1; d.Tvalid = a.Tvalid; d.alpha = a.alpha;
  if (a.d == 384) return launch_wide<384, 6>(d, stream);
  return launch_wide<512, 8>(d, stream);
}

}  // namespace fcenc
