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

}  // namespace fcenc
