// Fused macaron feed-forward block for gfx950 (bf16 MFMA, d_model = 256):
//
//     x <- x + alpha * (W2 silu(W1 LN(x) + b1) + b2)            (model.py:577-584, :663, :674)
//     optionally followed by the block's closing LayerNorm      (model.py:677)
//
// in ONE kernel: the (M, 4d) hidden activation never leaves the CU.  A workgroup (8 waves) owns
// 64 tokens.  It normalises them into an LDS tile, then walks the hidden dimension in chunks of
// 256 units: GEMM-1 (4 K-slices of W1) produces the chunk, bias + SiLU are applied on the
// accumulators and the bf16 chunk goes to a second LDS tile; GEMM-2 (4 K-slices of W2)
// accumulates it into the 256 x 64 output held in registers for the whole kernel.  Both weight
// matrices stream through one ring of three 32 KiB LDS stages as a single sequence of 32
// uniform slices (direct-to-LDS loads, counted vmcnt, one raw barrier per slice; see gemm.hip
// for the ring protocol and the source-side bank swizzle).  The epilogue transposes the output
// through LDS, adds the residual and, for the second FFN of a block, applies the closing
// LayerNorm on the full rows it owns.
//
// Per-workgroup traffic: 64 x 256 fp32 in, the same out, and 1 MiB of weights from L2.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int D = 256;            // model width this kernel is built for
constexpr int BM = 64;            // tokens per workgroup
constexpr int HC = 256;           // hidden units per chunk
constexpr int NTHREADS = 512;
constexpr int A_BYTES = BM * D * 2;       // 32 KiB normalised input tile
constexpr int H_BYTES = BM * HC * 2;      // 32 KiB hidden chunk tile
constexpr int STAGE = 32768;              // one weight slice: 256 rows x 64 bf16
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = A_BYTES + H_BYTES + NSTAGE * STAGE;   // 160 KiB

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// [64 tokens][256] bf16 tiles (512-byte rows): 16-byte chunk c of row m at c ^ (m & 15);
// conflict-free for the 16-row fragment reads although every row starts on bank 0
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

struct FfnDev {
  float* X;                 // [M][256] fp32 residual stream, updated in place
  const float* ln_g;        // pre-LayerNorm affine
  const float* ln_b;
  const bf16* W1;           // [FF][256]
  const float* b1;          // [FF]
  const bf16* W2;           // [256][FF]
  const float* b2;          // [256]
  const float* post_g;      // closing LayerNorm affine or null
  const float* post_b;
  bf16* Xc;                 // optional bf16 copy of the new X
  float* compact;           // optional [B][Tvalid][256] copy of the new X without padding frames
  int M, FF, Tp, Tvalid;
  float alpha;
};

__global__ __launch_bounds__(NTHREADS) void ffn_fused_kernel(const FfnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* a_tile = smem;
  char* h_tile = smem + A_BYTES;
  char* ring = smem + A_BYTES + H_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as a scalar: everything derived from it (ring destinations, weight row groups) stays
  // in SGPRs and the direct-to-LDS loads need no per-lane address arithmetic
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int nchunk = p.FF / HC;
  const int nslice = nchunk * 8;

  // ---- weight loader: wave w stages rows 32w..32w+31 of every slice (4 wave-instructions).
  // Addresses are a wave-uniform base (slice and row group, scalar ALU) plus a per-lane 32-bit
  // offset fixed for the whole kernel, so issuing a slice costs no vector arithmetic.
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const uint32_t w1off = (uint32_t)((srow * D + schunk * 8) * 2);
  const uint32_t w2off = (uint32_t)((srow * p.FF + schunk * 8) * 2);
  const char* w1base = reinterpret_cast<const char*>(p.W1) + (size_t)(32 * wave) * D * 2;
  const char* w2base = reinterpret_cast<const char*>(p.W2) + (size_t)(32 * wave) * p.FF * 2;
  const size_t w1q = (size_t)8 * D * 2, w2q = (size_t)8 * p.FF * 2;
  auto issue = [&](int sl_raw) {
    const int sl = sl_raw < nslice ? sl_raw : nslice - 1;
    const int c = sl >> 3, j = sl & 7;
    char* dst = ring + (sl_raw % NSTAGE) * STAGE + wave * 4096;
    if (j < 4) {
      const char* s = w1base + ((size_t)c * HC * D + j * 64) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(s + q * w1q + w1off, dst + q * 1024);
    } else {
      const char* s = w2base + ((size_t)c * HC + (j - 4) * 64) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(s + q * w2q + w2off, dst + q * 1024);
    }
  };
  issue(0);
  issue(1);      // slice 2 follows from the first gate

  // ---- pre-LayerNorm of this workgroup's 64 tokens into a_tile.  Wave w owns tokens 8w..8w+7
  // in two passes of four: DPP row g (16 lanes) takes token 8w + 4*pass + g and lane r of the row
  // features 4r + 64k (k = 0..3), so the row sums are four DPP adds and no LDS traffic.  The same
  // mapping is used by the epilogue, so the fp32 input stays in registers for the residual add.
  f32x4 rows[2][4];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int m = m0 + wave * 8 + 4 * ps + g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rows[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < p.M) rows[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
    }
  }
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

  f32x4 acc2[2][4];      // output: features 32w + 16i + 4g.., tokens 16j + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[2][4];      // hidden chunk: units 32w + 16i + 4g.., tokens 16j + r
  f32x4 bias1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  u32x4 wf[2][2] = {}, tf[2][4] = {};    // weight / token-side fragments (8 bf16 each), double-buffered by step parity

  // Fragment reads are issued by inline asm with precomputed per-lane LDS addresses: the compiler
  // would wait lgkmcnt(0) right after issuing them (it cannot count LDS returns across the
  // direct-to-LDS loads), which serialises LDS latency with the MFMAs.  Here the reads of step t+1
  // are issued first and a COUNTED lgkmcnt(6) waits only for the six fragments of step t.
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t woff[2][2], toff[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      woff[a][ks] = lds0 + A_BYTES + H_BYTES + (uint32_t)(32 * wave + 16 * a + r) * 128u +
                    (uint32_t)((((ks << 2) | g) ^ (r & 7)) << 4);                       // a = fragment i
      toff[a][ks] = lds0 + (uint32_t)r * 512u + (uint32_t)((((a << 3) | (ks << 2) | g) ^ r) << 4);   // a = slice parity
    }

  // gate(sl): slice sl has landed in every wave's part of the ring and every wave is done reading
  // the stage that slice sl+2 will overwrite.  vmcnt(4): only the youngest slice (4 loads per wave)
  // may stay in flight; lgkmcnt(0): this wave's own LDS reads and writes (a_tile, h_tile) are complete.
  auto gate = [&](int sl) {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(sl + 2);
  };

  // Software pipeline over k32 steps (16 per chunk; step t = slice t>>1 of the chunk, half t&1):
  // slices 0-3 are GEMM-1 against a_tile, 4-7 GEMM-2 against h_tile.  The only bubble is the
  // GEMM-1 -> GEMM-2 hand-over (step 7 -> 8), where the hidden chunk is written and published first.
  auto load_step = [&](auto tc, int sl) {
    constexpr int t = decltype(tc)::value;
    constexpr int j = t >> 1, ks = t & 1, jq = j & 3, buf = t & 1;
    constexpr int timm = (j < 4 ? 0 : A_BYTES) + (jq >> 1) * 256;
    const uint32_t stage = (uint32_t)(sl % NSTAGE) * STAGE;
    lds_read128<0>(wf[buf][0], woff[0][ks] + stage);
    lds_read128<0>(wf[buf][1], woff[1][ks] + stage);
    lds_read128<timm>(tf[buf][0], toff[jq & 1][ks]);
    lds_read128<timm + 8192>(tf[buf][1], toff[jq & 1][ks]);
    lds_read128<timm + 16384>(tf[buf][2], toff[jq & 1][ks]);
    lds_read128<timm + 24576>(tf[buf][3], toff[jq & 1][ks]);
  };
  auto mma_step = [&](auto tc, int pending) {
    constexpr int t = decltype(tc)::value;
    constexpr int buf = t & 1;
    // the six fragments of this step are older than the `pending` ones just issued for the next
    if (pending)
      asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(tf[buf][0]), "+v"(tf[buf][1]),
                   "+v"(tf[buf][2]), "+v"(tf[buf][3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(tf[buf][0]), "+v"(tf[buf][1]),
                   "+v"(tf[buf][2]), "+v"(tf[buf][3]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if constexpr (t < 8) mma(acc1[i][jj], __builtin_bit_cast(bf16x8, wf[buf][i]), __builtin_bit_cast(bf16x8, tf[buf][jj]));
        else mma(acc2[i][jj], __builtin_bit_cast(bf16x8, wf[buf][i]), __builtin_bit_cast(bf16x8, tf[buf][jj]));
      }
  };

  gate(0);
  load_step(std::integral_constant<int, 0>{}, 0);
  for (int c = 0; c < nchunk; ++c) {
    static_for<0, 16>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      int pending = 0;
      if constexpr (t == 3) {
        // this chunk's bias (used after step 7), by inline asm so that the compiler, which cannot
        // count vmcnt across the direct-to-LDS loads, does not drain them; the gate of slice 3
        // (step 5) retires it: it is older than the slice that gate leaves in flight.
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* bp = p.b1 + c * HC + 32 * wave + 16 * i + 4 * g;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias1[i]) : "v"(bp) : "memory");
        }
      }
      if constexpr (t < 15 && t != 7) {
        constexpr int nt = t + 1;
        if constexpr ((nt & 1) == 0) gate(8 * c + (nt >> 1));
        if constexpr (t == 5) asm volatile("" : "+v"(bias1[0]), "+v"(bias1[1]));   // bias valid from here on
        load_step(std::integral_constant<int, nt>{}, 8 * c + (nt >> 1));
        pending = 1;
      } else if constexpr (t == 15) {
        if (c + 1 < nchunk) {
          gate(8 * (c + 1));
          load_step(std::integral_constant<int, 0>{}, 8 * (c + 1));
          pending = 1;
        }
      }
      if constexpr (t == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc1[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      mma_step(tc, pending);
      if constexpr (t == 7) {
        // bias + SiLU on the accumulators, bf16 chunk to h_tile[token][unit], then GEMM-2 may start
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int u = 32 * wave + 16 * i + 4 * g;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 v = acc1[i][jj] + bias1[i];
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16)silu_f(v[k]);
            *reinterpret_cast<bf16x4*>(h_tile + wide_off(16 * jj + r, u >> 3) + 2 * (u & 7)) = o;
          }
        }
        gate(8 * c + 4);
        load_step(std::integral_constant<int, 8>{}, 8 * c + 4);
      }
    });
  }

  // ---- epilogue: out tile [64 tokens][256 features] fp32 through LDS (ring area, 64 KiB) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* ct = reinterpret_cast<float*>(ring);
  auto cidx = [](int row, int col) { return row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3)); };
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      *reinterpret_cast<f32x4*>(ct + cidx(16 * jj + r, 32 * wave + 16 * i + 4 * g)) = acc2[i][jj];
  __syncthreads();

  // same token / feature ownership as the prologue: 256 contiguous bytes per 16-lane row and access
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = wave * 8 + 4 * ps + g;
    const int m = m0 + row;
    f32x4 y[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r + 64 * k;
      y[k] = *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
      y[k] = rows[ps][k] + (y[k] + *reinterpret_cast<const f32x4*>(p.b2 + col)) * p.alpha;
      s += y[k][0] + y[k][1] + y[k][2] + y[k][3];
    }
    if (p.post_g) {
      const float mean = row16_sum(s) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y[k] -= mean;
        q += y[k][0] * y[k][0] + y[k][1] * y[k][1] + y[k][2] * y[k][2] + y[k][3] * y[k][3];
      }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / D) + 1e-5f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        y[k] = y[k] * rstd * *reinterpret_cast<const f32x4*>(p.post_g + 4 * r + 64 * k) +
               *reinterpret_cast<const f32x4*>(p.post_b + 4 * r + 64 * k);
    }
    if (m >= p.M) continue;
    const int bt = m / p.Tp, tt = m - bt * p.Tp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = 4 * r + 64 * k;
      *reinterpret_cast<f32x4*>(p.X + (size_t)m * D + col) = y[k];
      if (p.Xc) store4(p.Xc + (size_t)m * D + col, y[k]);
      if (p.compact && tt < p.Tvalid)
        *reinterpret_cast<f32x4*>(p.compact + ((size_t)bt * p.Tvalid + tt) * D + col) = y[k];
    }
  }
}

}  // namespace

bool ffn_fused_supported(int dtype, int d, int ff) { return dtype == DT_BF16 && d == D && ff % HC == 0 && ff >= HC; }

int launch_ffn_fused(const FfnArgs& a, hipStream_t stream) {
  if (!ffn_fused_supported(DT_BF16, a.d, a.ff)) return fail("ffn_fused: unsupported shape d=%d ff=%d", a.d, a.ff);
  if (a.M <= 0) return fail("ffn_fused: empty input");
  if (!a.X || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.W2 || !a.b2) return fail("ffn_fused: null operand");
  if ((a.post_g == nullptr) != (a.post_b == nullptr)) return fail("ffn_fused: closing LayerNorm affine incomplete");
  if (a.compact && (a.Tp <= 0 || a.Tvalid <= 0 || a.Tvalid > a.Tp || a.M % a.Tp)) return fail("ffn_fused: compact geometry");
  FfnDev d{};
  d.X = a.X; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const bf16*)a.W1; d.b1 = a.b1; d.W2 = (const bf16*)a.W2;
  d.b2 = a.b2; d.post_g = a.post_g; d.post_b = a.post_b; d.Xc = (bf16*)a.Xc; d.compact = a.compact;
  d.M = a.M; d.FF = a.ff; d.Tp = a.Tp > 0 ? a.Tp : 1; d.Tvalid = a.Tvalid; d.alpha = a.alpha;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          LDS_BYTES) != hipSuccess)
    return fail("ffn_fused: cannot reserve %d bytes of LDS", LDS_BYTES);
  hipLaunchKernelGGL(ffn_fused_kernel, dim3((a.M + BM - 1) / BM), dim3(NTHREADS), LDS_BYTES, stream, d);
  return check_launch("ffn_fused");
}

}  // namespace fcenc
