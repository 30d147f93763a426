// Device-side body of the fused feed-forward block (see ffn.hip for the design), shared by the
// stand-alone kernel and by the kernel that runs it straight after the convolution module.
#pragma once
#include <type_traits>

#include "common.h"

namespace fcenc {
namespace ffnk {

constexpr int D = 256;            // model width this kernel is built for
constexpr int BM = 64;            // tokens per workgroup
constexpr int HC = 256;           // hidden units per chunk
constexpr int NTHREADS = 512;
constexpr int A_BYTES = BM * D * 2;       // 32 KiB normalised input tile
constexpr int H_BYTES = BM * HC * 2;      // 32 KiB hidden chunk tile
constexpr int LDS_BYTES = A_BYTES + 2 * H_BYTES;   // two hidden tiles (chunks alternate); the fp32 output tile of the epilogue reuses the first 64 KiB
constexpr int DEPTH = 8;                  // k32 steps of weights in flight per wave
constexpr int CHUNK_VEC = 8 * 8 * 2 * 64; // 16-byte vectors per packed weight chunk (128 KiB)

// The block's parameter vectors that are used the moment they are read (LayerNorm affines, output and projection
// biases) live in LDS for the life of the block.  Vector loads return in issue order, and the CU's load path is one
// queue for all its waves, so such a vector read from global memory in the middle of the kernel waits for the
// 128 KiB of weights the workgroup has in flight in front of it (2 us, once per phase); read from LDS it waits for
// nothing.  (The hidden bias b1 is requested a whole GEMM ahead of its use and stays a global load.)  In floats:
template <int NQ>
struct Prm {
  static constexpr int LN_G = 0, LN_B = 256, B2 = 512, POST_G = 768, POST_B = 1024, BQ = 1280;
  static constexpr int FLOATS = BQ + NQ * 256;
  static constexpr int BYTES = FLOATS * 4;
  static constexpr int NV = (FLOATS / 4 + NTHREADS - 1) / NTHREADS;     // 16-byte vectors per thread
};
// LDS a kernel that runs the block on its own has to provide
template <int NQ>
constexpr int lds_bytes() { return LDS_BYTES + Prm<NQ>::BYTES; }

// [64 tokens][256] bf16 tiles (512-byte rows): 16-byte chunk c of row m at c ^ (m & 15);
// conflict-free for the 16-row fragment reads although every row starts on bank 0
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

struct FfnDev {
  float* X;                 // [M][256] fp32 residual stream, updated in place
  const float* ln_g;        // pre-LayerNorm affine
  const float* ln_b;
  const u32x4* W1;          // packed, see ffn_pack_kernel
  const float* b1;          // [FF]
  const u32x4* W2;          // packed
  const float* b2;          // [256]
  const float* post_g;      // closing LayerNorm affine or null
  const float* post_b;
  bf16* Xc;                 // optional bf16 copy of the new X
  float* compact;           // optional [B][Tvalid][256] copy of the new X without padding frames
  const u32x4* Wq;          // projection weights [256 * NQ][256], row-packed with paired fragments (qkv_pack); NQ > 0 only
  const float* bq;          // [256 * NQ]
  bf16* Q;                  // [M][256 * NQ] projection of the new X
  int M, Tp, Tvalid;
  float alpha;
  float* partial;           // split mode: [ff / 256][M][256] fp32 partial products of the second GEMM
};

// The parameter vectors on their way to LDS: `issue` starts the loads (ahead of the first weight fetches, so that
// they return first), `commit` writes them to the area; a barrier has to follow before anybody reads them.
template <int NQ>
struct PrmStage {
  using P = Prm<NQ>;
  f32x4 v[P::NV];
  __device__ __forceinline__ void issue(const FfnDev& p) {
#pragma unroll
    for (int i = 0; i < P::NV; ++i) {
      const int f = 4 * ((int)threadIdx.x + NTHREADS * i);     // segments are multiples of 256 floats: uniform per wave
      const float* src = nullptr;
      if (f < P::LN_B) src = p.ln_g + f;
      else if (f < P::B2) src = p.ln_b + (f - P::LN_B);
      else if (f < P::POST_G) src = p.b2 + (f - P::B2);
      else if (f < P::POST_B) src = p.post_g ? p.post_g + (f - P::POST_G) : nullptr;
      else if (f < P::BQ) src = p.post_b ? p.post_b + (f - P::POST_B) : nullptr;
      else if (f < P::FLOATS) src = p.bq + (f - P::BQ);
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (src) v[i] = *reinterpret_cast<const f32x4*>(src);
    }
  }
  __device__ __forceinline__ void commit(float* area) const {
#pragma unroll
    for (int i = 0; i < P::NV; ++i) {
      const int f = 4 * ((int)threadIdx.x + NTHREADS * i);
      if (f < P::FLOATS) *reinterpret_cast<f32x4*>(area + f) = v[i];
    }
  }
};

// One workgroup's share of the block: rows m0 .. m0+63.  `load_rows` fills the fp32 input rows in
// the ownership described below (a kernel reads them from p.X; a kernel that has just produced them
// hands them over in registers) and is called before the first weight fetches are issued.
// smem: lds_bytes<NQ>(), 16-byte aligned (LDS_BYTES when the caller has staged the parameters, see STAGED).
// Results go to p.X / p.Xc / p.compact (unless STORE is false) and p.Q; `emit` then receives the fp32 output rows, in the same ownership, for a caller that goes on
// with them.
// SPLIT: the workgroup takes ONE hidden chunk (blockIdx.y) and leaves its share of the second GEMM, without
// bias or residual, in p.partial for ffn_finish_kernel to add up (few tokens: more CUs share the weight stream).
// STAGED: the caller has put the parameter vectors (Prm<NQ>) at `prm` in LDS and a barrier behind them -- a kernel
// that runs several blocks requests all their parameters when it starts, ahead of every weight fetch.
template <int NCHUNK, int NQ, bool STORE, class LoadRows, class Emit, bool SPLIT = false, bool STAGED = false>
__device__ __forceinline__ void ffn_block(const FfnDev& p, char* smem, const int m0, LoadRows&& load_rows, Emit&& emit,
                                          const int stamp0 = -100, float* prm = nullptr) {
  static_assert(!SPLIT || (NCHUNK == 1 && NQ == 0 && !STORE), "split mode: one chunk, no projection, no direct store");
  using P = Prm<NQ>;
  const int c0 = SPLIT ? (int)blockIdx.y : 0;
  char* a_tile = smem;
  // Hidden chunks alternate between two tiles, so that a wave may write chunk c (bias + SiLU) while slower waves still
  // read chunk c - 1 in their second product: the one barrier per chunk is the one that publishes the chunk, and the
  // VALU work of the wave that is ahead overlaps the MFMAs of the wave it shares a SIMD with.  (Chunk c overwrites the
  // tile of chunk c - 2, which every wave has finished reading before it passed the barrier of chunk c - 1.)
  // (The second tile lies beyond the 16-bit offset field of an LDS instruction counted from the first address register;
  // its offset goes through an empty asm statement so that the compiler adds it where it is used instead of keeping a
  // second set of address registers alive, which spills.)
  auto h_tile_of = [&](int c) {
    uint32_t off = A_BYTES + (c & 1) * H_BYTES;
    if (c & 1) asm volatile("" : "+s"(off));
    return smem + off;
  };
  if constexpr (!STAGED) prm = reinterpret_cast<float*>(smem + LDS_BYTES);

  // thread coordinates behind an optimisation barrier: when this body follows other phases in one
  // kernel, the compiler must not keep addresses formed from the caller's copies alive across it
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr int NSTEP = NCHUNK * 16;      // per chunk: 8 steps of GEMM-1, then 8 of GEMM-2
  constexpr int NTOTAL = NSTEP + 8 * NQ;  // then 8 steps per 256-row block of the projection

  // ---- weight stream: step s reads two consecutive 1 KiB fragments of this wave's 16 KiB run ----
  const u32x4* w1s = p.W1 + (size_t)wave * 1024 + lane;
  const u32x4* w2s = p.W2 + (size_t)wave * 1024 + lane;
  u32x4 wr[DEPTH][2];
  auto fetch = [&](auto sc) {
    constexpr int s = decltype(sc)::value;
#ifdef FCENC_ABL_NOLOAD
    if constexpr (s >= DEPTH) return;      // timing experiment (wrong results): the ring is filled once and never again
#endif
    if constexpr (s < NSTEP) {
      constexpr int c = s >> 4, t = s & 15;
      const u32x4* src = (t < 8 ? w1s : w2s) + (size_t)(c + c0) * CHUNK_VEC + (t & 7) * 128;
      wr[s % DEPTH][0] = src[0];
      wr[s % DEPTH][1] = src[64];
    } else if constexpr (s < NTOTAL) {
      constexpr int q = (s - NSTEP) >> 3, ks = (s - NSTEP) & 7;
      const u32x4* src = p.Wq + (size_t)wave * 1024 + lane + (size_t)q * CHUNK_VEC + ks * 128;
      wr[s % DEPTH][0] = src[0];
      wr[s % DEPTH][1] = src[64];
    }
  };
  // what the first phase needs is requested first: the input rows, the parameter vectors, then the weights
  f32x4 rows[2][4];
  load_rows(rows);
  if constexpr (!STAGED) {
    PrmStage<NQ> stage;
    stage.issue(p);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
    __builtin_amdgcn_sched_barrier(0);
    stage.commit(prm);
    __syncthreads();
  } else {
    static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
  }
  FC_STAMP(smem, stamp0);

  // ---- pre-LayerNorm of this workgroup's 64 tokens into a_tile.  Wave w owns tokens 8w..8w+7
  // in two passes of four: DPP row g (16 lanes) takes token 8w + 4*pass + g and lane r of the row
  // features 4r + 64k (k = 0..3), so the row sums are four DPP adds and no LDS traffic.  The same
  // mapping is used by the epilogue, so the fp32 input stays in registers for the residual add.
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += rows[ps][k][0] + rows[ps][k][1] + rows[ps][k][2] + rows[ps][k][3];
    const float mean = row16_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 c = rows[ps][k] - mean;
      q += c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
    }
    const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(prm + P::LN_G + 4 * r + 64 * k);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(prm + P::LN_B + 4 * r + 64 * k);
      const f32x4 y = (rows[ps][k] - mean) * rstd * gg + bb;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)y[e];
      *reinterpret_cast<bf16x4*>(a_tile + wide_off(row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
    }
  }
  __syncthreads();
  FC_STAMP(smem, stamp0 + 1);

  f32x4 acc2[2][4];      // output: features 32w + 16i + 4g.., tokens 16j + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[2][4];      // hidden chunk: units 32w + 16i + 4g.., tokens 16j + r
  f32x4 bias1[2];

  // token-side fragments of step s: k32 step t&7 of a_tile (GEMM-1) or h_tile (GEMM-2)
  bf16x8 tf[2][4];
  auto tokens = [&](auto sc) {
    constexpr int s = decltype(sc)::value, t = s & 15;
#ifdef FCENC_ABL_NOLDS
    if constexpr (s >= 2) return;          // timing experiment (wrong results): the first two fragment sets are reused
#endif
    if constexpr (s < NSTEP) {
      const char* tile = t < 8 ? a_tile : h_tile_of(s >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        tf[s & 1][j] = *reinterpret_cast<const bf16x8*>(tile + wide_off(16 * j + r, 4 * (t & 7) + g));
    }
  };

  // One step = the next step's token fragments (LDS), this step's 8 MFMAs, then the weight fetch
  // DEPTH steps ahead into the registers the MFMAs just released.  The scheduling fences keep the
  // compiler from sinking the fetches towards their use, which would shorten the prefetch distance;
  // it still places the counted vmcnt / lgkmcnt waits itself.
  tokens(std::integral_constant<int, 0>{});
  static_for<0, NSTEP>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int c = s >> 4, t = s & 15;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bias1[i] = *reinterpret_cast<const f32x4*>(p.b1 + (c + c0) * HC + 32 * wave + 16 * i + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (t != 7) tokens(std::integral_constant<int, s + 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (t < 8) mma(acc1[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][i]), tf[s & 1][j]);
        else mma(acc2[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][i]), tf[s & 1][j]);
      }
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, s + DEPTH>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t == 7) {
      // bias + SiLU on the accumulators, bf16 chunk to its hidden tile [token][unit]; the barrier publishes the chunk
      FC_STAMP(smem, stamp0 + 2 + 2 * c);
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
          *reinterpret_cast<bf16x4*>(h_tile + wide_off(16 * j + r, u >> 3) + 2 * (u & 7)) = o;
        }
      }
      __syncthreads();
      FC_STAMP(smem, stamp0 + 3 + 2 * c);
      tokens(std::integral_constant<int, s + 1>{});
    }
  });
  FC_STAMP(smem, stamp0 + 2 + 2 * NCHUNK);

  // ---- epilogue: out tile [64 tokens][256 features] fp32 through LDS (both tiles, 64 KiB) ----
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
  auto cidx = [](int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * j + r, 32 * wave + 16 * i + 4 * g)) = acc2[i][j];
  __syncthreads();

  // same token / feature ownership as the prologue: 256 contiguous bytes per 16-lane row and access
  f32x4 y[2][4];
#ifdef FCENC_ABL_NOX
  const auto x_buf = buffer_of(nullptr, 0);                              // timing experiment: every store of X dropped
#else
  const auto x_buf = buffer_of(p.X, (uint32_t)p.M * (D * 4));
#endif
  const auto xc_buf = buffer_of(p.Xc, (uint32_t)p.M * (D * 2));
  const auto compact_buf = buffer_of(p.compact, (uint32_t)(p.M / p.Tp) * p.Tvalid * (D * 4));
  if constexpr (SPLIT) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = wave * 8 + 4 * ps + g;
      const int m = m0 + row;
      if (m >= p.M) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * r + 64 * k;
        *reinterpret_cast<f32x4*>(p.partial + ((size_t)c0 * p.M + m) * D + col) =
            *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
      }
    }
    return;
  }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g;
    const int m = m0 + row;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r + 64 * k;
      y[ps][k] = *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
      y[ps][k] = rows[ps][k] + (y[ps][k] + *reinterpret_cast<const f32x4*>(prm + P::B2 + col)) * p.alpha;
      s += y[ps][k][0] + y[ps][k][1] + y[ps][k][2] + y[ps][k][3];
    }
    if (p.post_g) {
      const float mean = row16_sum(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y[ps][k] -= mean;
        q += y[ps][k][0] * y[ps][k][0] + y[ps][k][1] * y[ps][k][1] + y[ps][k][2] * y[ps][k][2] + y[ps][k][3] * y[ps][k][3];
      }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        y[ps][k] = y[ps][k] * rstd * *reinterpret_cast<const f32x4*>(prm + P::POST_G + 4 * r + 64 * k) +
                   *reinterpret_cast<const f32x4*>(prm + P::POST_B + 4 * r + 64 * k);
    }
    if constexpr (STORE) {
      // rows beyond M, the copies a caller did not ask for and the padding frames of the compact copy are dropped
      // by the buffers' range check: no branch, so the waits behind these stores are counted exactly
      const int bt = m / p.Tp, tt = m - bt * p.Tp;
      const uint32_t xo = (uint32_t)m * (D * 4);
      const uint32_t co = tt < p.Tvalid ? (uint32_t)(bt * p.Tvalid + tt) * (D * 4) : BUF_SKIP;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * r + 64 * k;
        buffer_store4(x_buf, xo + 4 * col, y[ps][k]);
        buffer_store4_bf16(xc_buf, (xo >> 1) + 2 * col, y[ps][k]);
        buffer_store4(compact_buf, co + 4 * col, y[ps][k]);
      }
    }
  }
  emit(y);
  FC_STAMP(smem, stamp0 + 3 + 2 * NCHUNK);

  if constexpr (NQ > 0) {
    // ---- projection of the new rows: bf16 tile over a_tile, NQ blocks of 256 output features ----
    __syncthreads();                 // every thread has read its part of the fp32 output tile
    // lane coordinates from the lane counter: copies kept for this phase across the chunk loop would be spilled
    const int lane_q = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int g = lane_q >> 4, r = lane_q & 15;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = wave * 8 + 4 * ps + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)y[ps][k][e];
        *reinterpret_cast<bf16x4*>(a_tile + wide_off(row, (r >> 1) + 8 * k) + 8 * (r & 1)) = o;
      }
    }
    __syncthreads();
    auto qtokens = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
#ifdef FCENC_ABL_NOLDS
      if constexpr (s >= NSTEP + 2) return;
#endif
      if constexpr (s < NTOTAL) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          tf[s & 1][j] = *reinterpret_cast<const bf16x8*>(a_tile + wide_off(16 * j + r, 4 * ((s - NSTEP) & 7) + g));
      }
    };
    f32x4 acc3[2][4];
#ifdef FCENC_ABL_NOQ
    const auto q_buf = buffer_of(nullptr, 0);                            // timing experiment: every store of Q dropped
#else
    const auto q_buf = buffer_of(p.Q, (uint32_t)p.M * (512 * NQ));      // rows beyond M are dropped by its range check
#endif
    qtokens(std::integral_constant<int, NSTEP>{});
    static_for<NSTEP, NTOTAL>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int q = (s - NSTEP) >> 3, ks = (s - NSTEP) & 7;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      qtokens(std::integral_constant<int, s + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc3[i][j], __builtin_bit_cast(bf16x8, wr[s % DEPTH][i]), tf[s & 1][j]);
      __builtin_amdgcn_sched_barrier(0);
      fetch(std::integral_constant<int, s + DEPTH>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks == 7) {
        // + bias, straight from the accumulators to Q.  The projection matrix is packed with its fragment rows
        // paired (qkv_pack), so fragment i of a lane holds features 32 wave + 8g + 4i .. + 3 of token 16j + r: the
        // two fragments are one 16-byte store, the four lanes that share a token 64 contiguous bytes of its row.
        // No LDS round trip.
        {
          const int u = 32 * wave + 8 * g;
          const f32x4 bb0 = *reinterpret_cast<const f32x4*>(prm + P::BQ + 256 * q + u);
          const f32x4 bb1 = *reinterpret_cast<const f32x4*>(prm + P::BQ + 256 * q + u + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            buffer_store8_bf16(q_buf, (uint32_t)(m0 + 16 * j + r) * (512 * NQ) + 2 * (256 * q + u), acc3[0][j] + bb0,
                               acc3[1][j] + bb1);
        }
        if constexpr (q + 1 < NQ) FC_STAMP(smem, stamp0 + 5 + 2 * NCHUNK + q);
      }
    });
    FC_STAMP(smem, stamp0 + 4 + 2 * NCHUNK);
  }
}


template <int NCHUNK, int NQ, class LoadRows>
__device__ __forceinline__ void ffn_block(const FfnDev& p, char* smem, const int m0, LoadRows&& load_rows,
                                          const int stamp0 = -100) {
  ffn_block<NCHUNK, NQ, true>(p, smem, m0, load_rows, [](f32x4 (&)[2][4]) {}, stamp0);
}

}  // namespace ffnk
}  // namespace fcenc
