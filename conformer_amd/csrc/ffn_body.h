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
constexpr int LDS_BYTES = A_BYTES + H_BYTES;   // the fp32 output tile of the epilogue reuses both
constexpr int DEPTH = 8;                  // k32 steps of weights in flight per wave
constexpr int CHUNK_VEC = 8 * 8 * 2 * 64; // 16-byte vectors per packed weight chunk (128 KiB)

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
  const u32x4* Wq;          // projection weights [256 * NQ][256], row-packed (rows_pack_kernel); NQ > 0 only
  const float* bq;          // [256 * NQ]
  bf16* Q;                  // [M][256 * NQ] projection of the new X
  int M, Tp, Tvalid;
  float alpha;
  float* partial;           // split mode: [ff / 256][M][256] fp32 partial products of the second GEMM
};

// One workgroup's share of the block: rows m0 .. m0+63.  `load_rows` fills the fp32 input rows in
// the ownership described below (a kernel reads them from p.X; a kernel that has just produced them
// hands them over in registers) and is called after the first weight fetches have been issued.
// smem: LDS_BYTES, 16-byte aligned.  Results go to p.X / p.Xc / p.compact (unless STORE is false) and
// p.Q; `emit` then receives the fp32 output rows, in the same ownership, for a caller that goes on
// with them.
// SPLIT: the workgroup takes ONE hidden chunk (blockIdx.y) and leaves its share of the second GEMM, without
// bias or residual, in p.partial for ffn_finish_kernel to add up (few tokens: more CUs share the weight stream).
template <int NCHUNK, int NQ, bool STORE, class LoadRows, class Emit, bool SPLIT = false>
__device__ __forceinline__ void ffn_block(const FfnDev& p, char* smem, const int m0, LoadRows&& load_rows, Emit&& emit,
                                          const int stamp0 = -100) {
  static_assert(!SPLIT || (NCHUNK == 1 && NQ == 0 && !STORE), "split mode: one chunk, no projection, no direct store");
  const int c0 = SPLIT ? (int)blockIdx.y : 0;
  char* a_tile = smem;
  char* h_tile = smem + A_BYTES;

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
  static_for<0, DEPTH>([&](auto sc) { fetch(sc); });
  FC_STAMP(smem, stamp0);

  // ---- pre-LayerNorm of this workgroup's 64 tokens into a_tile.  Wave w owns tokens 8w..8w+7
  // in two passes of four: DPP row g (16 lanes) takes token 8w + 4*pass + g and lane r of the row
  // features 4r + 64k (k = 0..3), so the row sums are four DPP adds and no LDS traffic.  The same
  // mapping is used by the epilogue, so the fp32 input stays in registers for the residual add.
  f32x4 rows[2][4];
  load_rows(rows);
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
      const f32x4 gg = *reinterpret_cast<const f32x4*>(p.ln_g + 4 * r + 64 * k);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * r + 64 * k);
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
      const char* tile = t < 8 ? a_tile : h_tile;
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
      // bias + SiLU on the accumulators, bf16 chunk to h_tile[token][unit]; the first barrier keeps
      // the previous chunk's GEMM-2 reads ahead of the overwrite, the second publishes the chunk
      FC_STAMP(smem, stamp0 + 2 + 2 * c);
      if constexpr (c > 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = 32 * wave + 16 * i + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc1[i][j] + bias1[i];
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (bf16)silu_f(v[k]);
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
      y[ps][k] = rows[ps][k] + (y[ps][k] + *reinterpret_cast<const f32x4*>(p.b2 + col)) * p.alpha;
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
        y[ps][k] = y[ps][k] * rstd * *reinterpret_cast<const f32x4*>(p.post_g + 4 * r + 64 * k) +
                   *reinterpret_cast<const f32x4*>(p.post_b + 4 * r + 64 * k);
    }
    if (!STORE || m >= p.M) continue;
    const int bt = m / p.Tp, tt = m - bt * p.Tp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r + 64 * k;
      *reinterpret_cast<f32x4*>(p.X + (size_t)m * D + col) = y[ps][k];
      if (p.Xc) store4(p.Xc + (size_t)m * D + col, y[ps][k]);
      if (p.compact && tt < p.Tvalid)
        *reinterpret_cast<f32x4*>(p.compact + ((size_t)bt * p.Tvalid + tt) * D + col) = y[ps][k];
    }
  }
  emit(y);
  FC_STAMP(smem, stamp0 + 3 + 2 * NCHUNK);

  if constexpr (NQ > 0) {
    // ---- projection of the new rows: bf16 tile over a_tile, NQ blocks of 256 output features ----
    __syncthreads();                 // every thread has read its part of the fp32 output tile
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
        // + bias, bf16 block to h_tile[token][feature], then 512 contiguous bytes per token row to Q
        if constexpr (q > 0) __syncthreads();      // the previous block has been copied out
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int u = 32 * wave + 16 * i + 4 * g;
          const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bq + 256 * q + u);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc3[i][j] + bb;
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16)v[k];
            *reinterpret_cast<bf16x4*>(h_tile + wide_off(16 * j + r, u >> 3) + 2 * (u & 7)) = o;
          }
        }
        __syncthreads();
        const int ch = tid & 31;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int row = (tid >> 5) + 16 * ps;
          const int m = m0 + row;
          const u32x4 v = *reinterpret_cast<const u32x4*>(h_tile + wide_off(row, ch));
          if (m < p.M) *reinterpret_cast<u32x4*>(p.Q + (size_t)m * (256 * NQ) + 256 * q + 8 * ch) = v;
        }
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
