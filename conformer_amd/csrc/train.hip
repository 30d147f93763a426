// Backward (and train-mode forward) kernels of the memory-bound pieces of a conformer block, fp32, for the
// training path (train.py:302-336 calls model.forward and loss.backward):
//   * LayerNorm backward (model.py:516, :578, :677 are nn.LayerNorm(d), eps 1e-5, affine)
//   * GLU over channels followed by the causal depthwise convolution, forward and backward
//     (model.py:521-525; CausalConv1d model.py:236-249 with zero left padding, no cache in training).
//     BatchNorm in training mode needs statistics over the whole batch and stays outside these kernels.
// Every kernel is one pass over HBM with 16-byte accesses per lane; reductions over tokens (the affine /
// weight gradients) leave per-workgroup partial sums that the caller adds up, so results are deterministic.
#include "common.h"
#include "kernels.h"

namespace fcenc {

// ---------------------------------------------------------------------------------------------
// LayerNorm backward.  y = (x - mean) * rstd * g + b per row of d features:
//   xhat = (x - mean) * rstd,  a = dy * g
//   dx = rstd * (a - mean_d(a) - xhat * mean_d(a * xhat)),  dg = sum_rows dy * xhat,  db = sum_rows dy
// One wave per row at a time, NV float4 per lane (d <= 256 * NV); a workgroup of 4 waves walks rows
// blockIdx.x * 4 + wave, + 4 * gridDim.x, ... and keeps dg / db for its columns in registers, then adds
// the four waves through LDS and writes partial[blockIdx.x][2][d].
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            const float* __restrict__ g, float* __restrict__ dX,
                                                            float* __restrict__ partial, int M, int d) {
  __shared__ float red[4][2][NV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)d;
  f32x4 gg[NV], dg[NV], db[NV];
  bool on[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    on[i] = c < d;
    gg[i] = on[i] ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[i] = db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += 4 * gridDim.x) {
    f32x4 x[NV], dy[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const size_t at = (size_t)row * d + i * 256 + lane * 4;
      x[i] = on[i] ? *reinterpret_cast<const f32x4*>(X + at) : f32x4{0.f, 0.f, 0.f, 0.f};
      dy[i] = on[i] ? *reinterpret_cast<const f32x4*>(dY + at) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i])
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float c = x[i][k] - mean;
          q += c * c;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_d + 1e-5f);
    float sa = 0.f, sax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[i] = (x[i] - mean) * rstd;                // xhat (zero lanes stay harmless: dy and g are zero there)
      const f32x4 a = dy[i] * gg[i];
      sa += a[0] + a[1] + a[2] + a[3];
      sax += a[0] * x[i][0] + a[1] * x[i][1] + a[2] * x[i][2] + a[3] * x[i][3];
      dg[i] += dy[i] * x[i];
      db[i] += dy[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sa += __shfl_xor(sa, o);
      sax += __shfl_xor(sax, o);
    }
    const float ma = sa * inv_d, max_ = sax * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i])
        *reinterpret_cast<f32x4*>(dX + (size_t)row * d + i * 256 + lane * 4) = (dy[i] * gg[i] - ma - x[i] * max_) * rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<f32x4*>(&red[wave][0][i * 256 + lane * 4]) = dg[i];
    *reinterpret_cast<f32x4*>(&red[wave][1][i * 256 + lane * 4]) = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 256) {
    const int which = c / d, col = c - which * d;
    partial[((size_t)blockIdx.x * 2 + which) * d + col] =
        red[0][which][col] + red[1][which][col] + red[2][which][col] + red[3][which][col];
  }
}

int launch_layernorm_bwd(const float* X, const float* dY, const float* g, float* dX, float* partial, int nblocks, int M,
                         int d, hipStream_t stream) {
  if (!X || !dY || !g || !dX || !partial) return fail("layernorm_bwd: null operand");
  if (M <= 0 || d <= 0 || d % 4 || d > 1024) return fail("layernorm_bwd: M=%d d=%d (d %% 4 == 0, d <= 1024)", M, d);
  if (nblocks <= 0 || nblocks > 65535) return fail("layernorm_bwd: %d partial blocks", nblocks);
  if (((uintptr_t)X | (uintptr_t)dY | (uintptr_t)g | (uintptr_t)dX) & 15) return fail("layernorm_bwd: operands must be 16-byte aligned");
  const int nv = (d + 255) / 256;
  switch (nv) {
    case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nblocks), dim3(256), 0, stream, X, dY, g, dX, partial, M, d); break;
    case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nblocks), dim3(256), 0, stream, X, dY, g, dX, partial, M, d); break;
    case 3: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(nblocks), dim3(256), 0, stream, X, dY, g, dX, partial, M, d); break;
    default: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nblocks), dim3(256), 0, stream, X, dY, g, dX, partial, M, d); break;
  }
  return check_launch("layernorm_bwd");
}

// ---------------------------------------------------------------------------------------------
// GLU + causal depthwise convolution (kernel 9), token-major.
//   Y [B][T][2d]: pointwise1 output, value channels 0..d-1, gate channels d..2d-1 (nn.GLU(dim=channel))
//   G [B][T][d] = Y_value * sigmoid(Y_gate)
//   Z [B][T][d] = bias[c] + sum_{m=0..8} w9[m][c] * G[b][t-8+m][c],  G = 0 before the utterance starts
// A thread owns 4 channels and a run of GD_RUN frames, sliding a 9-frame register window down time;
// consecutive threads own consecutive channel groups, so every access is 16 bytes per lane along a row.
// ---------------------------------------------------------------------------------------------
constexpr int GD_RUN = 16;

__device__ __forceinline__ f32x4 glu4(const float* y, int d) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(y), gt = *reinterpret_cast<const f32x4*>(y + d);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = a[k] / (1.0f + expf(-gt[k]));
  return o;
}

__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ w9,
                                                             const float* __restrict__ bias, float* __restrict__ G,
                                                             float* __restrict__ Z, int T, int d) {
  const int cg = blockIdx.x * 256 + threadIdx.x;      // channel group
  const int c = cg * 4;
  if (c >= d) return;
  const int b = blockIdx.z, t0 = blockIdx.y * GD_RUN;
  f32x4 w[9];
#pragma unroll
  for (int m = 0; m < 9; ++m) w[m] = *reinterpret_cast<const f32x4*>(w9 + (size_t)m * d + c);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + c);
  const float* yb = Y + (size_t)b * T * 2 * d + c;
  f32x4 win[9];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int t = t0 - 8 + m;
    win[m] = t >= 0 ? glu4(yb + (size_t)t * 2 * d, d) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < GD_RUN; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    win[8] = glu4(yb + (size_t)t * 2 * d, d);
    f32x4 z = bb;
#pragma unroll
    for (int m = 0; m < 9; ++m) z += w[m] * win[m];
    const size_t at = ((size_t)b * T + t) * d + c;
    *reinterpret_cast<f32x4*>(G + at) = win[8];
    *reinterpret_cast<f32x4*>(Z + at) = z;
#pragma unroll
    for (int m = 0; m < 8; ++m) win[m] = win[m + 1];
  }
}

int launch_glu_dwconv_fwd(const float* Y, const float* w9, const float* bias, float* G, float* Z, int B, int T, int d,
                          hipStream_t stream) {
  if (!Y || !w9 || !bias || !G || !Z) return fail("glu_dwconv_fwd: null operand");
  if (B <= 0 || T <= 0 || d <= 0 || d % 4 || B > 65535) return fail("glu_dwconv_fwd: B=%d T=%d d=%d", B, T, d);
  if (((uintptr_t)Y | (uintptr_t)w9 | (uintptr_t)bias | (uintptr_t)G | (uintptr_t)Z) & 15) return fail("glu_dwconv_fwd: operands must be 16-byte aligned");
  const int runs = (T + GD_RUN - 1) / GD_RUN;
  if (runs > 65535) return fail("glu_dwconv_fwd: utterance too long");
  hipLaunchKernelGGL(glu_dwconv_fwd_kernel, dim3((d / 4 + 255) / 256, runs, B), dim3(256), 0, stream, Y, w9, bias, G, Z, T, d);
  return check_launch("glu_dwconv_fwd");
}

// Backward.  With dZ the gradient of Z:
//   dG[b][t][c] = sum_{m=0..8} w9[m][c] * dZ[b][t+8-m][c]           (terms with t+8-m >= T drop out)
//   dY_value = dG * sigmoid(gate),  dY_gate = dG * value * sigmoid(gate) * (1 - sigmoid(gate))
//   dw9[m][c] = sum_{b,t} dZ[b][t][c] * G[b][t-8+m][c],  dbias[c] = sum_{b,t} dZ[b][t][c]
// Same ownership as the forward kernel.  A thread walks its run once: frame t of the run receives dG from a
// 9-frame window of dZ that looks AHEAD (t .. t+8), and contributes dZ[t] * G[t-8+m] to dw9 from a window of G
// that looks back.  The weight / bias gradients of a thread go to partial[run block][10][d] (m = 0..8, then the
// bias row); the caller adds the blocks up.
__global__ __launch_bounds__(256) void glu_dwconv_bwd_kernel(const float* __restrict__ Y, const float* __restrict__ G,
                                                             const float* __restrict__ dZ, const float* __restrict__ w9,
                                                             float* __restrict__ dY, float* __restrict__ partial, int T,
                                                             int d) {
  const int cg = blockIdx.x * 256 + threadIdx.x;
  const int c = cg * 4;
  if (c >= d) return;
  const int b = blockIdx.z, t0 = blockIdx.y * GD_RUN;
  f32x4 w[9], dw[9], dbias = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 9; ++m) {
    w[m] = *reinterpret_cast<const f32x4*>(w9 + (size_t)m * d + c);
    dw[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const size_t base = (size_t)b * T * d + c;
  auto load = [&](const float* p, int t) {
    return (t >= 0 && t < T) ? *reinterpret_cast<const f32x4*>(p + base + (size_t)t * d) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  f32x4 gwin[9], zwin[9];          // gwin[m] = G[t-8+m], zwin[j] = dZ[t+j]
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    gwin[m] = load(G, t0 - 8 + m);
    zwin[m + 1] = load(dZ, t0 + m);
  }
#pragma unroll
  for (int i = 0; i < GD_RUN; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
#pragma unroll
    for (int j = 0; j < 8; ++j) zwin[j] = zwin[j + 1];
    zwin[8] = load(dZ, t + 8);
    gwin[8] = load(G, t);
    f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      dg += w[m] * zwin[8 - m];          // dZ[t + 8 - m]
      dw[m] += zwin[0] * gwin[m];        // dZ[t] * G[t - 8 + m]
    }
    dbias += zwin[0];
    const float* yp = Y + ((size_t)b * T + t) * 2 * d + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(yp), gt = *reinterpret_cast<const f32x4*>(yp + d);
    f32x4 da, dgt;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s = 1.0f / (1.0f + expf(-gt[k]));
      da[k] = dg[k] * s;
      dgt[k] = dg[k] * a[k] * s * (1.0f - s);
    }
    float* dp = dY + ((size_t)b * T + t) * 2 * d + c;
    *reinterpret_cast<f32x4*>(dp) = da;
    *reinterpret_cast<f32x4*>(dp + d) = dgt;
#pragma unroll
    for (int m = 0; m < 8; ++m) gwin[m] = gwin[m + 1];
  }
  float* pp = partial + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * 10) * d + c;
#pragma unroll
  for (int m = 0; m < 9; ++m) *reinterpret_cast<f32x4*>(pp + (size_t)m * d) = dw[m];
  *reinterpret_cast<f32x4*>(pp + (size_t)9 * d) = dbias;
}

int glu_dwconv_partial_blocks(int B, int T) { return B * ((T + GD_RUN - 1) / GD_RUN); }

int launch_glu_dwconv_bwd(const float* Y, const float* G, const float* dZ, const float* w9, float* dY, float* partial,
                          int B, int T, int d, hipStream_t stream) {
  if (!Y || !G || !dZ || !w9 || !dY || !partial) return fail("glu_dwconv_bwd: null operand");
  if (B <= 0 || T <= 0 || d <= 0 || d % 4 || B > 65535) return fail("glu_dwconv_bwd: B=%d T=%d d=%d", B, T, d);
  if (((uintptr_t)Y | (uintptr_t)G | (uintptr_t)dZ | (uintptr_t)w9 | (uintptr_t)dY | (uintptr_t)partial) & 15)
    return fail("glu_dwconv_bwd: operands must be 16-byte aligned");
  const int runs = (T + GD_RUN - 1) / GD_RUN;
  if (runs > 65535) return fail("glu_dwconv_bwd: utterance too long");
  hipLaunchKernelGGL(glu_dwconv_bwd_kernel, dim3((d / 4 + 255) / 256, runs, B), dim3(256), 0, stream, Y, G, dZ, w9, dY,
                     partial, T, d);
  return check_launch("glu_dwconv_bwd");
}

// ---------------------------------------------------------------------------------------------
// Backward of nn.Linear in one call (model.py:348-350, 376, 429, 520, 541, 579-582, 751 are the Linear /
// 1x1 Conv1d layers whose gradients train.py:329 asks for):
//   dX [M][K] = dY W          GEMM with A = dY [M][N],    W-operand = W^T  [K][N]
//   dW [N][K] = dY^T X        GEMM with A = dY^T [N][M'], W-operand = X^T  [K][M']   (M' = M rounded up to 64, zeros)
//   db [N]    = column sums of dY
// The GEMM kernel reads both operands along the contraction index, so the three transposed operands are made
// here, by one tiled pass each that also converts to the compute type and writes the zero padding; dY's pass
// leaves the straight bf16 copy the dX product needs as well.  Everything is enqueued by one host call.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dstT,
                                                             T* __restrict__ dstS, int R, int C, int Rp) {
  // 64 x 64 tile: src rows r0.., columns c0..; dstT[c][r] for r < Rp (zeros beyond R), dstS[r][c] = src[r][c]
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tc = (threadIdx.x & 15) * 4, tr = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + tr + 16 * i;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      v = *reinterpret_cast<const f32x4*>(src + (size_t)r * C + c0 + tc);
      if (dstS) store4(dstS + (size_t)r * C + c0 + tc, v);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[tc + k][tr + 16 * i] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tr + 16 * i;
    store4(dstT + (size_t)(c0 + c) * Rp + r0 + tc, f32x4{tile[c][tc], tile[c][tc + 1], tile[c][tc + 2], tile[c][tc + 3]});
  }
}

// column sums of src [M][N] in two ordered steps: workgroup (x, y) adds rows rpb*y .. rpb*y+rpb-1 (rpb = 256) of columns 64x .. 64x+63
// (16 row lanes x 16 float4 columns, lanes merged in a fixed order) into out[y][N]; a second launch over those
// partial rows leaves the total.  Deterministic.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, float* __restrict__ out, int M, int N, int rpb) {
  __shared__ f32x4 red[16][16];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cq * 4;
  const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rl; r < r1; r += 16) s += *reinterpret_cast<const f32x4*>(src + (size_t)r * N + c);
  red[rl][cq] = s;
  __syncthreads();
  if (rl == 0) {
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][cq];
    *reinterpret_cast<f32x4*>(out + (size_t)blockIdx.y * N + c) = s;
  }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// out[i] = sum over the `planes` slices of part[s][i], s ascending (the planes of a cut contraction)
__global__ __launch_bounds__(256) void sum_planes_kernel(const float* __restrict__ part, float* __restrict__ out, int planes,
                                                         size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)256 * gridDim.x) {
    f32x4 s = *reinterpret_cast<const f32x4*>(part + i * 4);
    for (int k = 1; k < planes; ++k) s += *reinterpret_cast<const f32x4*>(part + ((size_t)k * n4 + i) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = s;
  }
}

// The weight gradient dW [N][K] contracts over the M tokens: few output tiles, a long contraction.  It is cut into
// slices that run side by side (about two workgroups per CU in all) and are added up in order afterwards.
struct DwPlan {
  int splits, chunk, mp;      // slices, tokens per slice (multiple of 64), padded token count = splits * chunk
};
static DwPlan dw_plan(int M, int N, int K) {
  const int tiles = (N / 128) * (K / 128);
  int s = (512 + tiles - 1) / tiles;
  const int most = (M + 255) / 256;            // at least 256 tokens per slice
  if (s > most) s = most;
  if (s < 1) s = 1;
  DwPlan p;
  p.chunk = ((M + s - 1) / s + 63) / 64 * 64;
  p.splits = (M + p.chunk - 1) / p.chunk;
  p.mp = p.splits * p.chunk;
  return p;
}

size_t linear_bwd_workspace_bytes(int dtype, int M, int N, int K) {
  const size_t es = dtype == DT_BF16 ? 2 : 4;
  const DwPlan pl = dw_plan(M, N, K);
  const size_t mp = (size_t)pl.mp;
  size_t b = align256((size_t)N * mp * es) + align256((size_t)K * mp * es) + align256((size_t)K * N * es);
  if (pl.splits > 1) b += align256((size_t)pl.splits * N * K * sizeof(float));
  if (dtype == DT_BF16) b += align256((size_t)M * N * es);
  b += align256((size_t)((M + 255) / 256) * N * sizeof(float));     // partial column sums of dY
  return b;   // (the planes of the cut dW contraction, counted above, lie behind them)
}

template <typename T>
static void transpose_launch(const float* src, T* dstT, T* dstS, int R, int C, int Rp, hipStream_t stream) {
  hipLaunchKernelGGL(transpose_cast_kernel<T>, dim3(C / 64, Rp / 64), dim3(256), 0, stream, src, dstT, dstS, R, C, Rp);
}

int launch_linear_bwd(int dtype, const float* X, const float* dY, const float* W, float* dX, float* dW, float* db,
                      void* ws, int M, int N, int K, hipStream_t stream) {
  if (dtype != DT_BF16 && dtype != DT_F32) return fail("linear_bwd: dtype %d", dtype);
  if (!dY || !ws) return fail("linear_bwd: null operand");
  if (M <= 0 || N % 128 || K % 128 || N <= 0 || K <= 0) return fail("linear_bwd: M=%d N=%d K=%d (N, K multiples of 128)", M, N, K);
  if ((dX && !W) || (dW && !X)) return fail("linear_bwd: dX needs W, dW needs X");
  if (((uintptr_t)X | (uintptr_t)dY | (uintptr_t)W | (uintptr_t)dX | (uintptr_t)dW | (uintptr_t)db | (uintptr_t)ws) & 15)
    return fail("linear_bwd: operands must be 16-byte aligned");
  const bool bf = dtype == DT_BF16;
  const size_t es = bf ? 2 : 4;
  const DwPlan pl = dw_plan(M, N, K);
  const int mp = pl.mp;
  char* p = static_cast<char*>(ws);
  char* dyT = p;  p += align256((size_t)N * mp * es);
  char* xT = p;   p += align256((size_t)K * mp * es);
  char* wT = p;   p += align256((size_t)K * N * es);
  char* dyS = bf ? p : nullptr;
  if (bf) p += align256((size_t)M * N * es);
  float* colpart = reinterpret_cast<float*>(p);
  p += align256((size_t)((M + 255) / 256) * N * sizeof(float));
  float* planes = reinterpret_cast<float*>(p);
  const bool need_dyS = bf && dX;
  if (dW || need_dyS) {
    if (bf) transpose_launch<bf16>(dY, (bf16*)dyT, need_dyS ? (bf16*)dyS : nullptr, M, N, mp, stream);
    else transpose_launch<float>(dY, (float*)dyT, nullptr, M, N, mp, stream);
  }
  if (dX) {
    if (bf) transpose_launch<bf16>(W, (bf16*)wT, nullptr, N, K, N, stream);
    else transpose_launch<float>(W, (float*)wT, nullptr, N, K, N, stream);
    GemmArgs a;
    a.A = bf ? (const void*)dyS : (const void*)dY; a.W = wT; a.X = dX;
    a.M = M; a.N = K; a.K = N; a.lda = N; a.ldo = K; a.alpha = 1.f; a.beta = 0.f;
    if (int rc = launch_gemm(dtype, EPI_RESID, a, stream)) return rc;
  }
  if (dW) {
    if (bf) transpose_launch<bf16>(X, (bf16*)xT, nullptr, M, K, mp, stream);
    else transpose_launch<float>(X, (float*)xT, nullptr, M, K, mp, stream);
    GemmArgs a;
    a.A = dyT; a.W = xT; a.X = pl.splits > 1 ? planes : dW;
    a.M = N; a.N = K; a.K = pl.chunk; a.lda = mp; a.ldw = mp; a.splits = pl.splits; a.ldo = K; a.alpha = 1.f; a.beta = 0.f;
    if (int rc = launch_gemm(dtype, EPI_RESID, a, stream)) return rc;
    if (pl.splits > 1) {
      const size_t n4 = (size_t)N * K / 4;
      hipLaunchKernelGGL(sum_planes_kernel, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256), 0,
                         stream, planes, dW, pl.splits, n4);
    }
  }
  if (db) {
    const int rb = (M + 255) / 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(N / 64, rb), dim3(256), 0, stream, dY, rb > 1 ? colpart : db, M, N, 256);
    if (rb > 1) hipLaunchKernelGGL(colsum_kernel, dim3(N / 64, 1), dim3(256), 0, stream, colpart, db, rb, N, rb);
  }
  return check_launch("linear_bwd");
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d with batch statistics + SiLU, token-major (model.py:528-537: self.norm then self.activation in
// training mode; the reference normalises over batch x time of every channel, padding frames included).
//   forward : mean_c, var_c (biased) over the M rows; y = silu((z - mean) * rstd * g + b);
//             running_mean / running_var <- (1 - momentum) * old + momentum * {mean, var * M / (M - 1)}
//   backward: u = zhat * g + b, s = sigmoid(u), du = dy * s * (1 + u * (1 - s));
//             db = sum du, dg = sum du * zhat, dz = g * rstd * (du - db / M - zhat * dg / M)
// Statistics: each workgroup takes BN_ROWS rows and leaves {sum, sum of squares about its own mean} per channel;
// one workgroup then merges the parts in order with the pairwise-update formula in fp64 (no cancellation,
// deterministic).  The backward sums are merged the same way.
// ---------------------------------------------------------------------------------------------
constexpr int BN_ROWS = 64;

__global__ __launch_bounds__(256) void bn_part_kernel(const float* __restrict__ Z, float* __restrict__ part, int M, int d) {
  // thread = (row lane rl of 256 / dq, float4 column cq); partial sums over the lane's rows, lanes merged through LDS
  extern __shared__ float sm[];
  const int dq = d >> 2, lanes = 256 / dq;
  const int cq = threadIdx.x % dq, rl = threadIdx.x / dq;
  const int r0 = blockIdx.x * BN_ROWS, r1 = min(M, r0 + BN_ROWS);
  f32x4* red = reinterpret_cast<f32x4*>(sm);                       // [lanes][dq]
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (rl < lanes)
    for (int r = r0 + rl; r < r1; r += lanes) s += *reinterpret_cast<const f32x4*>(Z + (size_t)r * d + cq * 4);
  if (rl < lanes) red[rl * dq + cq] = s;
  __syncthreads();
  f32x4 tot = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < lanes; ++i) tot += red[i * dq + cq];
  const f32x4 mean = tot * (1.0f / (float)(r1 - r0));
  __syncthreads();
  f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
  if (rl < lanes)
    for (int r = r0 + rl; r < r1; r += lanes) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(Z + (size_t)r * d + cq * 4) - mean;
      q += c * c;
    }
  if (rl < lanes) red[rl * dq + cq] = q;
  __syncthreads();
  if (rl == 0) {
    f32x4 m2 = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < lanes; ++i) m2 += red[i * dq + cq];
    float* o = part + (size_t)blockIdx.x * 2 * d + cq * 4;
    *reinterpret_cast<f32x4*>(o) = tot;
    *reinterpret_cast<f32x4*>(o + d) = m2;
  }
}

__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ part, int nblk, int M, int d,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ run_mean, float* __restrict__ run_var,
                                                        float momentum, float eps, float* __restrict__ stat) {
  // stat [4][d]: mean, rstd, scale = g * rstd, shift = b - mean * scale
  for (int c = blockIdx.x * 256 + threadIdx.x; c < d; c += 256 * gridDim.x) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const double nb = (double)min(BN_ROWS, M - k * BN_ROWS);
      const double mb = (double)part[(size_t)k * 2 * d + c] / nb, qb = (double)part[(size_t)k * 2 * d + d + c];
      const double delta = mb - mean, nn = n + nb;
      m2 += qb + delta * delta * n * nb / nn;
      mean += delta * nb / nn;
      n = nn;
    }
    const double var = m2 / n;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = g[c] * rstd;
    stat[c] = (float)mean;
    stat[d + c] = rstd;
    stat[2 * d + c] = scale;
    stat[3 * d + c] = b[c] - (float)mean * scale;
    if (run_mean) {
      run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
      run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(n > 1.0 ? m2 / (n - 1.0) : var);
    }
  }
}

__global__ __launch_bounds__(256) void bn_silu_apply_kernel(const float* __restrict__ Z, const float* __restrict__ stat,
                                                            float* __restrict__ Y, size_t total4, int d) {
  const int dq = d >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)256 * gridDim.x) {
    const int c = (int)(i % dq) * 4;
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + i * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(stat + 2 * d + c), sh = *reinterpret_cast<const f32x4*>(stat + 3 * d + c);
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float u = z[k] * sc[k] + sh[k];
      y[k] = u / (1.0f + expf(-u));
    }
    *reinterpret_cast<f32x4*>(Y + i * 4) = y;
  }
}

__device__ __forceinline__ void bn_du(const f32x4& z, const f32x4& dy, const f32x4& mean, const f32x4& rstd, const f32x4& g,
                                      const f32x4& b, f32x4& zhat, f32x4& du) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    zhat[k] = (z[k] - mean[k]) * rstd[k];
    const float u = zhat[k] * g[k] + b[k];
    const float s = 1.0f / (1.0f + expf(-u));
    du[k] = dy[k] * s * (1.0f + u * (1.0f - s));
  }
}

__global__ __launch_bounds__(256) void bn_silu_bwd_part_kernel(const float* __restrict__ Z, const float* __restrict__ dY,
                                                               const float* __restrict__ stat, const float* __restrict__ g,
                                                               const float* __restrict__ b, float* __restrict__ part, int M,
                                                               int d) {
  extern __shared__ float sm[];
  const int dq = d >> 2, lanes = 256 / dq;
  const int cq = threadIdx.x % dq, rl = threadIdx.x / dq;
  const int r0 = blockIdx.x * BN_ROWS, r1 = min(M, r0 + BN_ROWS);
  f32x4* red = reinterpret_cast<f32x4*>(sm);                       // [2][lanes][dq]
  f32x4 sb = f32x4{0.f, 0.f, 0.f, 0.f}, sg = sb;
  if (rl < lanes) {
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stat + cq * 4), rstd = *reinterpret_cast<const f32x4*>(stat + d + cq * 4);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + cq * 4), bb = *reinterpret_cast<const f32x4*>(b + cq * 4);
    for (int r = r0 + rl; r < r1; r += lanes) {
      f32x4 zhat, du;
      bn_du(*reinterpret_cast<const f32x4*>(Z + (size_t)r * d + cq * 4), *reinterpret_cast<const f32x4*>(dY + (size_t)r * d + cq * 4),
            mean, rstd, gg, bb, zhat, du);
      sb += du;
      sg += du * zhat;
    }
    red[rl * dq + cq] = sb;
    red[(lanes + rl) * dq + cq] = sg;
  }
  __syncthreads();
  if (rl == 0) {
    f32x4 tb = f32x4{0.f, 0.f, 0.f, 0.f}, tg = tb;
    for (int i = 0; i < lanes; ++i) {
      tb += red[i * dq + cq];
      tg += red[(lanes + i) * dq + cq];
    }
    float* o = part + (size_t)blockIdx.x * 2 * d + cq * 4;
    *reinterpret_cast<f32x4*>(o) = tb;
    *reinterpret_cast<f32x4*>(o + d) = tg;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ part, int nblk, int d,
                                                            float* __restrict__ dg, float* __restrict__ db) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < d; c += 256 * gridDim.x) {
    double sb = 0.0, sg = 0.0;
    for (int k = 0; k < nblk; ++k) {
      sb += (double)part[(size_t)k * 2 * d + c];
      sg += (double)part[(size_t)k * 2 * d + d + c];
    }
    db[c] = (float)sb;
    dg[c] = (float)sg;
  }
}

__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ dY,
                                                                const float* __restrict__ stat, const float* __restrict__ g,
                                                                const float* __restrict__ b, const float* __restrict__ dg,
                                                                const float* __restrict__ db, float* __restrict__ dZ,
                                                                size_t total4, int d, float inv_m) {
  const int dq = d >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)256 * gridDim.x) {
    const int c = (int)(i % dq) * 4;
    const f32x4 mean = *reinterpret_cast<const f32x4*>(stat + c), rstd = *reinterpret_cast<const f32x4*>(stat + d + c);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(b + c);
    const f32x4 sg = *reinterpret_cast<const f32x4*>(dg + c), sb = *reinterpret_cast<const f32x4*>(db + c);
    f32x4 zhat, du;
    bn_du(*reinterpret_cast<const f32x4*>(Z + i * 4), *reinterpret_cast<const f32x4*>(dY + i * 4), mean, rstd, gg, bb, zhat, du);
    *reinterpret_cast<f32x4*>(dZ + i * 4) = gg * rstd * (du - sb * inv_m - zhat * sg * inv_m);
  }
}

int bn_partial_blocks(int M) { return (M + BN_ROWS - 1) / BN_ROWS; }

static int bn_check(const char* what, int M, int d) {
  if (M <= 0 || d <= 0 || d % 4 || d > 1024) return fail("%s: M=%d d=%d (d %% 4 == 0, d <= 1024)", what, M, d);
  return 0;
}

int launch_bn_silu_train_fwd(const float* Z, const float* g, const float* b, float* run_mean, float* run_var,
                             float momentum, float eps, float* Y, float* stat, float* part, int M, int d,
                             hipStream_t stream) {
  if (!Z || !g || !b || !Y || !stat || !part) return fail("bn_silu_train_fwd: null operand");
  if (int rc = bn_check("bn_silu_train_fwd", M, d)) return rc;
  if ((run_mean == nullptr) != (run_var == nullptr)) return fail("bn_silu_train_fwd: running statistics come as a pair");
  if (((uintptr_t)Z | (uintptr_t)Y | (uintptr_t)stat | (uintptr_t)part) & 15) return fail("bn_silu_train_fwd: operands must be 16-byte aligned");
  const int nblk = bn_partial_blocks(M), dq = d / 4, lanes = 256 / dq;
  hipLaunchKernelGGL(bn_part_kernel, dim3(nblk), dim3(256), lanes * dq * 16, stream, Z, part, M, d);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, part, nblk, M, d, g, b, run_mean, run_var,
                     momentum, eps, stat);
  const size_t total4 = (size_t)M * dq;
  const int grid = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(grid), dim3(256), 0, stream, Z, stat, Y, total4, d);
  return check_launch("bn_silu_train_fwd");
}

int launch_bn_silu_train_bwd(const float* Z, const float* dY, const float* g, const float* b, const float* stat, float* dZ,
                             float* dg, float* db, float* part, int M, int d, hipStream_t stream) {
  if (!Z || !dY || !g || !b || !stat || !dZ || !dg || !db || !part) return fail("bn_silu_train_bwd: null operand");
  if (int rc = bn_check("bn_silu_train_bwd", M, d)) return rc;
  if (((uintptr_t)Z | (uintptr_t)dY | (uintptr_t)stat | (uintptr_t)dZ | (uintptr_t)dg | (uintptr_t)db | (uintptr_t)part | (uintptr_t)g | (uintptr_t)b) & 15)
    return fail("bn_silu_train_bwd: operands must be 16-byte aligned");
  const int nblk = bn_partial_blocks(M), dq = d / 4, lanes = 256 / dq;
  hipLaunchKernelGGL(bn_silu_bwd_part_kernel, dim3(nblk), dim3(256), 2 * lanes * dq * 16, stream, Z, dY, stat, g, b, part, M, d);
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, part, nblk, d, dg, db);
  const size_t total4 = (size_t)M * dq;
  const int grid = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(grid), dim3(256), 0, stream, Z, dY, stat, g, b, dg, db, dZ, total4, d,
                     1.0f / (float)M);
  return check_launch("bn_silu_train_bwd");
}

}  // namespace fcenc
