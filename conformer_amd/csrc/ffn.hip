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

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int nchunk = p.FF / HC;
  const int nslice = nchunk * 8;

  // ---- weight loader: wave w stages rows 32w..32w+31 of every slice (4 wave-instructions) ----
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const char* w1src = reinterpret_cast<const char*>(p.W1) + ((size_t)(32 * wave + srow) * D + schunk * 8) * 2;
  const char* w2src = reinterpret_cast<const char*>(p.W2) + ((size_t)(32 * wave + srow) * p.FF + schunk * 8) * 2;
  const size_t w1q = (size_t)8 * D * 2, w2q = (size_t)8 * p.FF * 2;
  auto issue = [&](int sl_raw) {
    const int sl = sl_raw < nslice ? sl_raw : nslice - 1;
    const int c = sl >> 3, j = sl & 7;
    char* dst = ring + (sl_raw % NSTAGE) * STAGE + wave * 4096;
    if (j < 4) {
      const char* s = w1src + ((size_t)c * HC * D + j * 64) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(s + q * w1q, dst + q * 1024);
    } else {
      const char* s = w2src + ((size_t)c * HC + (j - 4) * 64) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(s + q * w2q, dst + q * 1024);
    }
  };
  issue(0);
  issue(1);

  // ---- pre-LayerNorm of this workgroup's 64 tokens into a_tile (8 tokens per wave) ----
  {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(p.ln_g + 4 * lane);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.ln_b + 4 * lane);
    f32x4 rows[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {          // all eight row loads in flight before the first is used
      const int m = m0 + wave * 8 + t;
      rows[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < p.M) rows[t] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * lane);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = wave * 8 + t;
      const f32x4 v = rows[t];
      float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s * (1.0f / D);
      const f32x4 c = v - mean;
      float q = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
      const f32x4 y = c * rstd * gg + bb;
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (bf16)y[k];
      *reinterpret_cast<bf16x4*>(a_tile + wide_off(row, lane >> 1) + 8 * (lane & 1)) = o;
    }
  }

  f32x4 acc2[2][4];      // output: features 32w + 16i + 4g.., tokens 16j + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[2][4];      // hidden chunk: units 32w + 16i + 4g.., tokens 16j + r
  f32x4 bias1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  for (int c = 0; c < nchunk; ++c) {
   // the eight slices of a chunk are unrolled so that each has straight-line code: the compiler
   // can then count vmcnt across the direct-to-LDS loads instead of draining them
#pragma unroll
   for (int j = 0; j < 8; ++j) {
    const int sl = 8 * c + j;
    if (j == 2) {
      // this chunk's bias, needed after slice 3.  Loaded with inline asm one slice early: the
      // counted wait of slice 3 (below, with the registers as operands) retires it, and the compiler,
      // which cannot count across the direct-to-LDS loads, has no reason to drain them with vmcnt(0).
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float* bp = p.b1 + c * HC + 32 * wave + 16 * i + 4 * g;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias1[i]) : "v"(bp) : "memory");
      }
    }
    // slice sl landed (this wave's part); the younger slice may stay in flight.  lgkmcnt(0)
    // makes this wave's LDS writes (a_tile, h_tile) visible to the others past the barrier.
    if (j == 3)
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(bias1[0]), "+v"(bias1[1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(sl + 2);
    const char* wt = ring + (sl % NSTAGE) * STAGE;
    if (j < 4) {
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc1[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[2], af[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = lds_frag<bf16, 128>(wt, 32 * wave + 16 * i + r, ks, g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          af[jj] = *reinterpret_cast<const bf16x8*>(a_tile + wide_off(16 * jj + r, 8 * j + 4 * ks + g));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) mma(acc1[i][jj], wf[i], af[jj]);
      }
      if (j == 3) {
        // bias + SiLU on the accumulators, bf16 chunk to h_tile[token][unit]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int u = 32 * wave + 16 * i + 4 * g;
          const f32x4 bias = bias1[i];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 v = acc1[i][jj] + bias;
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16)silu_f(v[k]);
            *reinterpret_cast<bf16x4*>(h_tile + wide_off(16 * jj + r, u >> 3) + 2 * (u & 7)) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[2], hf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = lds_frag<bf16, 128>(wt, 32 * wave + 16 * i + r, ks, g);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          hf[jj] = *reinterpret_cast<const bf16x8*>(h_tile + wide_off(16 * jj + r, 8 * (j - 4) + 4 * ks + g));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) mma(acc2[i][jj], wf[i], hf[jj]);
      }
    }
   }
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

  // 32 lanes own one token row (8 features each); four rows per thread
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int cix = tid + NTHREADS * it;
    const int row = cix >> 5, col = (cix & 31) << 3;
    const int m = m0 + row;
    const bool live = m < p.M;
    f32x4 y0 = *reinterpret_cast<const f32x4*>(ct + cidx(row, col));
    f32x4 y1 = *reinterpret_cast<const f32x4*>(ct + cidx(row, col + 4));
    y0 = (y0 + *reinterpret_cast<const f32x4*>(p.b2 + col)) * p.alpha;
    y1 = (y1 + *reinterpret_cast<const f32x4*>(p.b2 + col + 4)) * p.alpha;
    float* xp = p.X + (size_t)(live ? m : 0) * D + col;
    if (live) {
      y0 += *reinterpret_cast<const f32x4*>(xp);
      y1 += *reinterpret_cast<const f32x4*>(xp + 4);
    }
    if (p.post_g) {
      float s = y0[0] + y0[1] + y0[2] + y0[3] + y1[0] + y1[1] + y1[2] + y1[3];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s * (1.0f / D);
      y0 -= mean;
      y1 -= mean;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) q += y0[k] * y0[k] + y1[k] * y1[k];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
      y0 = y0 * rstd * *reinterpret_cast<const f32x4*>(p.post_g + col) + *reinterpret_cast<const f32x4*>(p.post_b + col);
      y1 = y1 * rstd * *reinterpret_cast<const f32x4*>(p.post_g + col + 4) +
           *reinterpret_cast<const f32x4*>(p.post_b + col + 4);
    }
    if (!live) continue;
    *reinterpret_cast<f32x4*>(xp) = y0;
    *reinterpret_cast<f32x4*>(xp + 4) = y1;
    if (p.Xc) {
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (bf16)y0[k];
        o[4 + k] = (bf16)y1[k];
      }
      *reinterpret_cast<bf16x8*>(p.Xc + (size_t)m * D + col) = o;
    }
    if (p.compact) {
      const int b = m / p.Tp, t = m - b * p.Tp;
      if (t < p.Tvalid) {
        float* cp = p.compact + ((size_t)b * p.Tvalid + t) * D + col;
        *reinterpret_cast<f32x4*>(cp) = y0;
        *reinterpret_cast<f32x4*>(cp + 4) = y1;
      }
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
