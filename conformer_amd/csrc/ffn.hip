// Fused macaron feed-forward block for gfx950 (bf16 MFMA, d_model = 256):
//
//     x <- x + alpha * (W2 silu(W1 LN(x) + b1) + b2)            (model.py:577-584, :663, :674)
//     optionally followed by the block's closing LayerNorm      (model.py:677)
//
// in ONE kernel: the (M, 4d) hidden activation never leaves the CU.  A workgroup (8 waves) owns
// 64 tokens.  It normalises them into an LDS tile, then walks the hidden dimension in chunks of
// 256 units: GEMM-1 (8 k32 steps of W1) produces the chunk, bias + SiLU are applied on the
// accumulators and the bf16 chunk goes to a second LDS tile; GEMM-2 (8 k32 steps of W2)
// accumulates it into the 256 x 64 output held in registers for the whole kernel.
//
// Weights never touch LDS.  Wave w only ever multiplies rows 32w..32w+31 of a weight chunk, so it
// streams exactly those rows from L2 straight into MFMA A-fragments through a ring of registers,
// eight k32 steps (16 KiB per wave, 128 KiB per CU) ahead of their use.  With 64 tokens per CU
// the kernel's pace is the rate at which a CU can pull its 1 MiB of weights out of L2, and that
// rate is set by the bytes in flight; LDS could hold only half as many next to the two tiles.
// For the loads to be whole cache lines the weights are stored in fragment order (ffn_pack
// below): [chunk][wave][k32 step][fragment i][lane] x 16 bytes.
//
// Optionally the kernel goes on to project the new x (the attention's Q | K | V projection,
// model.py:366-368): the 64 updated rows are already on chip, so the projection only adds its
// weight stream and saves a kernel with its own read of x.
//
// Per-workgroup traffic: 64 x 256 fp32 in, the same out, and 1 MiB of weights from L2.
#include "ffn_body.h"
#include "kernels.h"

namespace fcenc {

namespace {

using namespace ffnk;

template <int NCHUNK, int NQ>
__global__ __launch_bounds__(NTHREADS) void ffn_fused_kernel(const FfnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int m0 = blockIdx.x * BM;
  ffn_block<NCHUNK, NQ>(p, smem, m0, [&](f32x4 (&rows)[2][4]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rows[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M) rows[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
      }
    }
  });
}

// Few tokens (a streaming chunk: 3 frames x 1024 streams = 48 token tiles): a workgroup per (token tile, hidden
// chunk), so 4x (8x) as many CUs share the weight stream that paces this kernel, each pulling a quarter (an
// eighth) of the weights; the partial products are added in a fixed order by ffn_finish_kernel.
__global__ __launch_bounds__(NTHREADS) void ffn_split_kernel(const FfnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int m0 = blockIdx.x * BM;
  auto load = [&](f32x4 (&rows)[2][4]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = m0 + wave * 8 + 4 * ps + g;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rows[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < p.M) rows[ps][k] = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + 4 * r + 64 * k);
      }
    }
  };
  auto none = [](f32x4 (&)[2][4]) {};
  ffn_block<1, 0, false, decltype(load)&, decltype(none)&, true>(p, smem, m0, load, none);
}

// x <- x + alpha * (sum of the partial products in chunk order + b2), optional closing LayerNorm, optional bf16
// copy and compact copy: one wave per token, four features per lane.
__global__ __launch_bounds__(256) void ffn_finish_kernel(const FfnDev p, int nchunk) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const int col = 4 * lane;
  f32x4 acc = *reinterpret_cast<const f32x4*>(p.partial + (size_t)m * D + col);
  for (int c = 1; c < nchunk; ++c) acc += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)c * p.M + m) * D + col);
  f32x4 y = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * D + col) +
            (acc + *reinterpret_cast<const f32x4*>(p.b2 + col)) * p.alpha;
  if (p.post_g) {
    float s = y[0] + y[1] + y[2] + y[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / D);
    y -= mean;
    float q = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / D) + 1e-5f);
    y = y * rstd * *reinterpret_cast<const f32x4*>(p.post_g + col) + *reinterpret_cast<const f32x4*>(p.post_b + col);
  }
  *reinterpret_cast<f32x4*>(p.X + (size_t)m * D + col) = y;
  if (p.Xc) store4(p.Xc + (size_t)m * D + col, y);
  if (p.compact) {
    const int bt = m / p.Tp, tt = m - bt * p.Tp;
    if (tt < p.Tvalid) *reinterpret_cast<f32x4*>(p.compact + ((size_t)bt * p.Tvalid + tt) * D + col) = y;
  }
}

// Row-major [R][256] (R a multiple of 256) -> fragment order by 256-row block: vector
// ((((c*8 + w)*8 + ks)*2 + i)*64 + lane) holds row 256c + 32w + 16i + r, columns 32ks + 8g .. (+8)
// PAIRED (the q | k | v projection, whose results leave straight from the accumulators): fragment row r of fragment i
// holds row 256c + 32w + 8(r >> 2) + 4i + (r & 3) instead, so that the two fragments of a wave give a lane (g, .) the
// eight consecutive output features 32w + 8g .. 8g + 7 of its token: one 16-byte store instead of two of 8.
template <bool PAIRED>
__global__ __launch_bounds__(256) void rows_pack_kernel(const u32x4* __restrict__ W, u32x4* __restrict__ Wp, int nvec) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  const int lane = v & 63, i = (v >> 6) & 1, ks = (v >> 7) & 7, w = (v >> 10) & 7, c = v >> 13;
  const int g = lane >> 4, r = lane & 15;
  const int row = PAIRED ? 8 * (r >> 2) + 4 * i + (r & 3) : 16 * i + r;
  Wp[v] = W[(size_t)(256 * c + 32 * w + row) * (D / 8) + 4 * ks + g];
}

// Row-major W1 [FF][256] and W2 [256][FF] -> fragment order.  Vector v of a packed matrix is
// (chunk c, wave w, k32 step ks, fragment i, lane 16g + r) and holds 8 consecutive K elements:
//   W1p: row 256c + 32w + 16i + r of W1, columns 32ks + 8g ..      (K = input features)
//   W2p: row        32w + 16i + r of W2, columns 256c + 32ks + 8g .. (K = hidden units of chunk c)
__global__ __launch_bounds__(256) void ffn_pack_kernel(const u32x4* __restrict__ W1, const u32x4* __restrict__ W2,
                                                       u32x4* __restrict__ W1p, u32x4* __restrict__ W2p, int ff) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= ff * D / 8) return;
  const int lane = v & 63, i = (v >> 6) & 1, ks = (v >> 7) & 7, w = (v >> 10) & 7, c = v >> 13;
  const int g = lane >> 4, r = lane & 15;
  const int row = 32 * w + 16 * i + r, k8 = 4 * ks + g;      // k8: column index in units of 8 elements
  W1p[v] = W1[(size_t)(256 * c + row) * (D / 8) + k8];
  W2p[v] = W2[(size_t)row * (ff / 8) + 32 * c + k8];
}

}  // namespace

static bool narrow(int dtype, int d, int ff) { return dtype == DT_BF16 && d == D && (ff == 1024 || ff == 2048); }

bool ffn_fused_supported(int dtype, int d, int ff) { return narrow(dtype, d, ff) || ffn_wide_supported(dtype, d, ff); }

// the split schedule exists for d = 256 only
bool ffn_split_pays(int M, int d, int ff) { return d == D && (long long)((M + BM - 1) / BM) * (ff / HC) <= 256; }

bool ffn_qkv_fused_supported(int dtype, int d, int ff, int heads) {
  return narrow(dtype, d, ff) && heads * 64 == 256;
}

int launch_ffn_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, hipStream_t stream) {
  if (!ffn_fused_supported(DT_BF16, d, ff)) return fail("ffn_pack: unsupported shape d=%d ff=%d", d, ff);
  if (!W1 || !W2 || !W1p || !W2p) return fail("ffn_pack: null operand");
  if (W1 == W1p || W2 == W2p) return fail("ffn_pack: packing cannot be done in place");
  if (d != D) return launch_ffn_wide_pack(W1, W2, W1p, W2p, d, ff, stream);
  const int nvec = ff * D / 8;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3((nvec + 255) / 256), dim3(256), 0, stream, (const u32x4*)W1, (const u32x4*)W2,
                     (u32x4*)W1p, (u32x4*)W2p, ff);
  return check_launch("ffn_pack");
}

int launch_qkv_pack(const void* W, void* Wp, int rows, int d, hipStream_t stream) {
  if (d != D || rows <= 0 || rows % 256) return fail("qkv_pack: unsupported shape [%d][%d]", rows, d);
  if (!W || !Wp || W == Wp) return fail("qkv_pack: null or aliased operand");
  const int nvec = rows * D / 8;
  hipLaunchKernelGGL(rows_pack_kernel<true>, dim3((nvec + 255) / 256), dim3(256), 0, stream, (const u32x4*)W, (u32x4*)Wp, nvec);
  return check_launch("qkv_pack");
}

int launch_rows_pack(const void* W, void* Wp, int rows, int d, hipStream_t stream) {
  if (d != D || rows <= 0 || rows % 256) return fail("rows_pack: unsupported shape [%d][%d]", rows, d);
  if (!W || !Wp || W == Wp) return fail("rows_pack: null or aliased operand");
  const int nvec = rows * D / 8;
  hipLaunchKernelGGL(rows_pack_kernel<false>, dim3((nvec + 255) / 256), dim3(256), 0, stream, (const u32x4*)W, (u32x4*)Wp, nvec);
  return check_launch("rows_pack");
}

int launch_ffn_fused(const FfnArgs& a, hipStream_t stream) {
  if (!ffn_fused_supported(DT_BF16, a.d, a.ff)) return fail("ffn_fused: unsupported shape d=%d ff=%d", a.d, a.ff);
  if (a.M <= 0) return fail("ffn_fused: empty input");
  if (a.M > (1 << 20)) return fail("ffn_fused: M=%d exceeds 2^20 tokens (32-bit buffer offsets)", a.M);
  if (!a.X || !a.ln_g || !a.ln_b || !a.W1 || !a.b1 || !a.W2 || !a.b2) return fail("ffn_fused: null operand");
  if ((a.post_g == nullptr) != (a.post_b == nullptr)) return fail("ffn_fused: closing LayerNorm affine incomplete");
  if (a.compact && (a.Tp <= 0 || a.Tvalid <= 0 || a.Tvalid > a.Tp || a.M % a.Tp)) return fail("ffn_fused: compact geometry");
  if (a.d != D) return launch_ffn_wide(a, stream);
  FfnDev d{};
  d.X = a.X; d.ln_g = a.ln_g; d.ln_b = a.ln_b; d.W1 = (const u32x4*)a.W1; d.b1 = a.b1; d.W2 = (const u32x4*)a.W2;
  d.b2 = a.b2; d.post_g = a.post_g; d.post_b = a.post_b; d.Xc = (bf16*)a.Xc; d.compact = a.compact;
  d.M = a.M; d.Tp = a.Tp > 0 ? a.Tp : 1; d.Tvalid = a.Tvalid; d.alpha = a.alpha;
  const dim3 grid((a.M + BM - 1) / BM), block(NTHREADS);
  auto launch = [&](auto kernel, dim3 g, int lds) {
    if (reserve_lds(reinterpret_cast<const void*>(kernel), lds) != hipSuccess)
      return fail("ffn_fused: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL(kernel, g, block, lds, stream, d);
    return 0;
  };
  if (a.partial) {
    if (a.Wq) return fail("ffn_fused: the split schedule does not carry the projection");
    d.partial = a.partial;
    const int nchunk = a.ff / HC;
    if (launch(ffn_split_kernel, dim3(grid.x, nchunk), lds_bytes<0>()) || check_launch("ffn_split")) return -1;
    hipLaunchKernelGGL(ffn_finish_kernel, dim3((a.M + 3) / 4), dim3(256), 0, stream, d, nchunk);
    return check_launch("ffn_finish");
  }
  if (a.Wq) {
    if (!a.bq || !a.Q || a.nq != 768) return fail("ffn_fused: the fused projection needs bias, output and 768 features (got %d)", a.nq);
    d.Wq = (const u32x4*)a.Wq; d.bq = a.bq; d.Q = (bf16*)a.Q;
    if (a.ff == 1024 ? launch(ffn_fused_kernel<4, 3>, grid, lds_bytes<3>()) : launch(ffn_fused_kernel<8, 3>, grid, lds_bytes<3>()))
      return -1;
  } else {
    if (a.ff == 1024 ? launch(ffn_fused_kernel<4, 0>, grid, lds_bytes<0>()) : launch(ffn_fused_kernel<8, 0>, grid, lds_bytes<0>()))
      return -1;
  }
  return check_launch("ffn_fused");
}

}  // namespace fcenc
