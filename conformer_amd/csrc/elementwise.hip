// Memory-bound kernels of the encoder: input log/standardisation statistics, the 1->C first
// subsampling convolution, LayerNorm (single or chained), the causal depthwise convolution with
// folded BatchNorm + SiLU, and the integer length / padding-mask kernel.  Each one is a single
// HBM pass with 16-byte accesses per lane.
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace fcenc {

// ---------------------------------------------------------------------------------------------
// input statistics: model.py:856-857 takes log(x + 1e-6) and standardises each utterance over
// (time, freq), padding included, with the UNBIASED std.  Pass 1 reduces sum / sum-of-squares of
// the log values in fp64 over `nsplit` slices per utterance; the consumer (conv1) finishes them.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_stats_kernel(const float* __restrict__ mel, double* __restrict__ partial,
                                                          int n_per_utt, int nsplit) {
  const int b = blockIdx.y, s = blockIdx.x;
  const long long chunk = ((n_per_utt / 4 + nsplit - 1) / nsplit) * 4ll;   // multiple of 4 elements
  const long long lo = (long long)s * chunk;
  const long long hi = lo + chunk < n_per_utt ? lo + chunk : n_per_utt;
  const float* base = mel + (size_t)b * n_per_utt;
  double sum = 0.0, sq = 0.0;
  for (long long i = lo + 4ll * threadIdx.x; i < hi; i += 4ll * 256) {
    if (i + 4 <= hi) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float l = logf(v[q] + 1e-6f);
        sum += (double)l;
        sq += (double)l * (double)l;
      }
    } else {
      for (long long k = i; k < hi; ++k) {
        const float l = logf(base[k] + 1e-6f);
        sum += (double)l;
        sq += (double)l * (double)l;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    sq += __shfl_xor(sq, o);
  }
  __shared__ double red[8];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 2] = sum;
    red[wave * 2 + 1] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < 4; ++w) {
      a += red[2 * w];
      c += red[2 * w + 1];
    }
    partial[((size_t)b * nsplit + s) * 2] = a;
    partial[((size_t)b * nsplit + s) * 2 + 1] = c;
  }
}

int launch_input_stats(const float* mel, double* partial, int B, int T, int F, int nsplit, hipStream_t stream) {
  if (B <= 0 || T <= 0 || F <= 0 || nsplit <= 0 || nsplit > 64) return fail("input_stats: bad shape");
  if (((long long)T * F) % 4) return fail("input_stats: T*F must be a multiple of 4");
  if ((long long)T * F >= (1ll << 31)) return fail("input_stats: utterance too long");
  hipLaunchKernelGGL(input_stats_kernel, dim3(nsplit, B), dim3(256), 0, stream, mel, partial, T * F, nsplit);
  return check_launch("input_stats");
}

// ---------------------------------------------------------------------------------------------
// conv1: Conv2d(1 -> C, 3x3, stride 2, pad 1) + BatchNorm2d(eval, folded) + SiLU (model.py:703-707)
// applied to the standardised log-mel, written channels-last [B][T1][F1][C].  One workgroup per
// output time row: the three input rows are standardised once into LDS, then every thread
// produces 8 channels of a pixel per step (16-byte stores, 32 lanes cover one pixel at C=256).
// Zero padding applies to the standardised tensor, so out-of-range taps contribute 0.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ mel, const double* __restrict__ partial,
                                                    int nsplit, const float* __restrict__ w9,
                                                    const float* __restrict__ bias, T* __restrict__ out, int Tin,
                                                    int F, int T1, int F1, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);        // [3][F + 2], column 0 is w = -1
  __shared__ float stat[2];
  const int b = blockIdx.y, t1 = blockIdx.x, tid = threadIdx.x;
  const int W = F + 2;
  if (tid < 64) {
    double s = 0.0, q = 0.0;
    if (tid < nsplit) {
      s = partial[((size_t)b * nsplit + tid) * 2];
      q = partial[((size_t)b * nsplit + tid) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    if (tid == 0) {
      const double n = (double)Tin * (double)F;
      const double mean = s / n;
      double var = (q - s * mean) / (n - 1.0);
      if (var < 0.0) var = 0.0;
      stat[0] = (float)mean;
      stat[1] = 1.0f / ((float)sqrt(var) + 1e-6f);
    }
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (int i = tid; i < 3 * W; i += 256) {
    const int kh = i / W, w = i - kh * W - 1;
    const int h = 2 * t1 - 1 + kh;
    float v = 0.f;
    if (h >= 0 && h < Tin && w >= 0 && w < F) v = (logf(mel[((size_t)b * Tin + h) * F + w] + 1e-6f) - mean) * rstd;
    rows[i] = v;
  }
  __syncthreads();
  const int groups = C >> 3;            // 8-channel groups
  const int cg = tid % groups, pl = tid / groups, npl = 256 / groups;
  float w[9][8], bb[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w9 + (size_t)k * C + cg * 8);
    const f32x4 c = *reinterpret_cast<const f32x4*>(w9 + (size_t)k * C + cg * 8 + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w[k][q] = a[q];
      w[k][4 + q] = c[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) bb[q] = bias[cg * 8 + q];
  T* orow = out + ((size_t)b * T1 + t1) * F1 * C;
  for (int f1 = pl; f1 < F1; f1 += npl) {
    float x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) x[kh * 3 + kw] = rows[kh * W + 2 * f1 + kw];   // column (2 f1 - 1 + kw) + 1
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float a = bb[q];
#pragma unroll
      for (int k = 0; k < 9; ++k) a = fmaf(w[k][q], x[k], a);
      y[q] = silu_f(a);
    }
    T* o = orow + (size_t)f1 * C + cg * 8;
    store4(o, f32x4{y[0], y[1], y[2], y[3]});
    store4(o + 4, f32x4{y[4], y[5], y[6], y[7]});
  }
}

// ---------------------------------------------------------------------------------------------
// conv1 on the matrix cores (bf16 output, C = 32 * NP <= 256 channels).  The 3x3 taps of a pixel
// are a 9-element dot product per channel, far too short for MFMA on its own, but one k32 step
// has room for the taps three times over, which buys fp32 accuracy from bf16 operands: with
// w = wh + wl and x = xh + xl (bf16 head and bf16 remainder),
//   k  0..7   wh[0..7] * xh[0..7]        k  8..15  wh[0..7] * xl[0..7]      k 16..23  wl[0..7] * xh[0..7]
//   k 24..28  wh[8]*xh[8], wh[8]*xl[8], wl[8]*xh[8], bh * 1, bl * 1          (bias = bh + bl)
// so a lane group g = k / 8 supplies one kind of operand and the dropped wl * xl terms are below
// 2^-16 relative.  Weights are the A operand (16 channels per fragment, kept in registers for the
// whole kernel), 16 pixels the B operand, and fragment row 4g'+q of the pair (2P, 2P+1) stands
// for channel 32P + 8g' + 4*half + q, so that a lane ends up with 8 consecutive channels of one
// pixel: one 16-byte store.  A workgroup standardises 2*RT+1 input rows into LDS once and
// produces RT output rows; wave w takes pixel groups w, w+8, ...
// ---------------------------------------------------------------------------------------------
// Output rows per workgroup: the set-up of a workgroup (input rows, statistics, two barriers, weight fragments) issues
// no stores, and the kernel is paced by its store stream, so the more rows a workgroup covers the fewer gaps the
// stream has (same shape: 320 us at 16 rows, 305 at 32, 285 at 64) -- as long as the grid still fills the chip twice.
// Input elements each thread stages: (2 RT + 1) (F + 2) <= 512 NPRE.
template <int RT>
struct C1Geo {
  static constexpr int NPRE = RT == 16 ? 6 : RT == 32 ? 11 : 21;      // F <= 91 / 84 / 81
};

__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
  hi = (bf16)v;
  lo = (bf16)(v - (float)hi);
}

template <int NP, int C1_RT>
__global__ __launch_bounds__(512) void conv1_mfma_kernel(const float* __restrict__ mel, const double* __restrict__ partial,
                                                         int nsplit, const float* __restrict__ w9,
                                                         const float* __restrict__ bias, bf16* __restrict__ out,
                                                         int Tin, int F, int T1, int F1) {
  constexpr int C = 32 * NP;
  constexpr int NR = 2 * C1_RT + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* rows = reinterpret_cast<float*>(smem);        // [NR][F + 2], column 0 is w = -1
  __shared__ float stat[2];
  // Output transposition, 2 KiB per wave: [16 pixels][128 bytes], 16-byte chunk c of pixel x at chunk c ^ (x & 7).
  // The accumulators give a lane 16 bytes of a pixel per pair of MFMAs and four lanes 64 contiguous bytes, half a
  // cache line; stores of half lines reach 3.9 TB/s, stores of whole lines 5 TB/s and more.  So two pairs of MFMAs
  // (64 channels) go through this tile and come back as whole 128-byte lines, eight pixels per store instruction.
  __shared__ __attribute__((aligned(16))) char xpose_all[8 * 2048];
  const int b = blockIdx.y, t1_0 = blockIdx.x * C1_RT, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int W = F + 2;
  char* xpose = xpose_all + 2048 * wave;

  // raw input first (independent of the statistics), standardised once those are known
  constexpr int NPRE = C1Geo<C1_RT>::NPRE;
  float raw[NPRE];
  uint32_t inside = 0;                    // bit i: element i is a real input sample, not zero padding
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = tid + 512 * i;
    const int kh = e / W, w = e - kh * W - 1;
    const int hrow = 2 * t1_0 - 1 + kh;
    raw[i] = 1.f;
    if (kh < NR && hrow >= 0 && hrow < Tin && w >= 0 && w < F) {
      raw[i] = mel[((size_t)b * Tin + hrow) * F + w];
      inside |= 1u << i;
    }
  }
  if (tid < 64) {
    double s = 0.0, q = 0.0;
    if (tid < nsplit) {
      s = partial[((size_t)b * nsplit + tid) * 2];
      q = partial[((size_t)b * nsplit + tid) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    if (tid == 0) {
      const double n = (double)Tin * (double)F;
      const double mean = s / n;
      double var = (q - s * mean) / (n - 1.0);
      if (var < 0.0) var = 0.0;
      stat[0] = (float)mean;
      stat[1] = 1.0f / ((float)sqrt(var) + 1e-6f);
    }
  }

  // weight fragments: af[2P + half], lane (g, r) holds 8 K slots of fragment row r
  bf16x8 af[2 * NP];
#pragma unroll
  for (int f = 0; f < 2 * NP; ++f) {
    const int ch = 32 * (f >> 1) + 8 * (r >> 2) + 4 * (f & 1) + (r & 3);
    bf16x8 a;
    if (g < 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bf16 hi, lo;
        split_bf16(w9[(size_t)k * C + ch], hi, lo);
        a[k] = g == 2 ? lo : hi;
      }
    } else {
      bf16 wh, wl, bh, bl;
      split_bf16(w9[(size_t)8 * C + ch], wh, wl);
      split_bf16(bias[ch], bh, bl);
      a[0] = wh; a[1] = wh; a[2] = wl; a[3] = bh; a[4] = bl;
      a[5] = a[6] = a[7] = (bf16)0.f;
    }
    af[f] = a;
  }

  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = tid + 512 * i;
    if (e < NR * W) rows[e] = ((inside >> i) & 1u) ? (logf(raw[i] + 1e-6f) - mean) * rstd : 0.f;
  }
  __syncthreads();

  const int nrow = T1 - t1_0 < C1_RT ? T1 - t1_0 : C1_RT;
  const int npix = nrow * F1;
  bf16* obase = out + ((size_t)b * T1 + t1_0) * F1 * C + 8 * g;
  for (int grp = wave; 16 * grp < npix; grp += 8) {
    const int pix = 16 * grp + r;
    const bool live = pix < npix;
    const int pc = live ? pix : 0;
    const int tl = pc / F1, f1 = pc - tl * F1;
    const float* x0 = rows + (2 * tl) * W + 2 * f1;     // tap (kh, kw) at x0[kh * W + kw]
    bf16x8 xf;
    if (g < 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bf16 hi, lo;
        split_bf16(x0[(k / 3) * W + (k % 3)], hi, lo);
        xf[k] = g == 1 ? lo : hi;
      }
    } else {
      bf16 hi, lo;
      split_bf16(x0[2 * W + 2], hi, lo);
      xf[0] = hi; xf[1] = lo; xf[2] = hi; xf[3] = (bf16)1.f; xf[4] = (bf16)1.f;
      xf[5] = xf[6] = xf[7] = (bf16)0.f;
    }
    // pixel (lane >> 3) + 8 * half of the group and chunk lane & 7 of its line: what this lane stores after the tile
    const int sx = lane >> 3, sc = lane & 7;
    const bool live_a = 16 * grp + sx < npix, live_b = 16 * grp + sx + 8 < npix;
    bf16* oa = obase - 8 * g + (size_t)(16 * grp + sx) * C + 8 * sc;
#pragma unroll
    for (int P = 0; P < NP; ++P) {
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
      mma(a0, af[2 * P], xf);
      mma(a1, af[2 * P + 1], xf);
      bf16x8 y;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#ifdef FCENC_C1_NOSILU
        y[q] = (bf16)a0[q];               // timing experiment (wrong results): no activation
        y[4 + q] = (bf16)a1[q];
#else
        y[q] = (bf16)silu_f(a0[q]);
        y[4 + q] = (bf16)silu_f(a1[q]);
#endif
      }
      // channels 32P + 8g .. +7 of pixel r = chunk 4 (P & 1) + g of the pixel's line 64 (P >> 1) .. +63
      *reinterpret_cast<bf16x8*>(xpose + 128 * r + 16 * ((4 * (P & 1) + g) ^ (r & 7))) = y;
      if (P & 1) {
        const bf16x8 ya = *reinterpret_cast<const bf16x8*>(xpose + 128 * sx + 16 * (sc ^ sx));
        const bf16x8 yb = *reinterpret_cast<const bf16x8*>(xpose + 128 * (sx + 8) + 16 * (sc ^ sx));
#ifdef FCENC_C1_NOSTORE
        if (live_a && __builtin_bit_cast(unsigned short, ya[0]) == 0x7fc1) *reinterpret_cast<bf16x8*>(oa + 64 * (P >> 1)) = ya;   // timing experiment
        if (live_b && __builtin_bit_cast(unsigned short, yb[0]) == 0x7fc1) *reinterpret_cast<bf16x8*>(oa + 8 * C + 64 * (P >> 1)) = yb;
#else
        if (live_a) *reinterpret_cast<bf16x8*>(oa + 64 * (P >> 1)) = ya;
        if (live_b) *reinterpret_cast<bf16x8*>(oa + 8 * C + 64 * (P >> 1)) = yb;
#endif
      }
    }
  }
}

int launch_conv1(int dtype, const float* mel, const double* partial, int nsplit, const float* w9, const float* bias,
                 void* out, int B, int T, int F, int C, hipStream_t stream) {
  if (B <= 0 || T <= 0 || F <= 0) return fail("conv1: empty input");
  if (dtype == DT_BF16 && C == 256 && 33 * (F + 2) <= 512 * C1Geo<16>::NPRE && nsplit > 0 && nsplit <= 64) {
    const int T1 = (T + 1) / 2, F1 = (F + 1) / 2;
    auto launch = [&](auto kernel, int rt) {
      const size_t lds = (size_t)(2 * rt + 1) * (F + 2) * sizeof(float);
      hipLaunchKernelGGL(kernel, dim3((T1 + rt - 1) / rt, B), dim3(512), lds, stream, mel, partial, nsplit, w9, bias,
                         (bf16*)out, T, F, T1, F1);
    };
    // the largest row count that still gives every CU two workgroups twice over and whose staging fits
    auto fits = [&](int rt, int npre) { return (long long)((T1 + rt - 1) / rt) * B >= 1024 && (2 * rt + 1) * (F + 2) <= 512 * npre; };
    if (fits(64, C1Geo<64>::NPRE)) launch(conv1_mfma_kernel<8, 64>, 64);
    else if (fits(32, C1Geo<32>::NPRE)) launch(conv1_mfma_kernel<8, 32>, 32);
    else launch(conv1_mfma_kernel<8, 16>, 16);
    return check_launch("conv1");
  }
  if (C % 8 || 256 % (C / 8) || C > 2048) return fail("conv1: C=%d must be 8 * a divisor of 256", C);
  if (nsplit <= 0 || nsplit > 64) return fail("conv1: nsplit=%d", nsplit);
  const int T1 = (T + 1) / 2, F1 = (F + 1) / 2;
  const size_t lds = (size_t)3 * (F + 2) * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(conv1_kernel<bf16>, dim3(T1, B), dim3(256), lds, stream, mel, partial, nsplit, w9, bias,
                       (bf16*)out, T, F, T1, F1, C);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(conv1_kernel<float>, dim3(T1, B), dim3(256), lds, stream, mel, partial, nsplit, w9, bias,
                       (float*)out, T, F, T1, F1, C);
  else
    return fail("conv1: dtype %d", dtype);
  return check_launch("conv1");
}

// ---------------------------------------------------------------------------------------------
// Statistics + conv1 for callers that cannot allocate (code under HIP graph capture, serving loops,
// harnesses that forbid hipMalloc): the partial sums go through a fixed array that is part of the
// code object instead of a caller-provided buffer.  Every GPU the library runs on holds its own copy
// of the array, at its own address, until the process ends.
// ---------------------------------------------------------------------------------------------
constexpr int kStaticMaxBatch = 4096;     // utterances per call
constexpr int kStaticNsplit = 16;         // slices per utterance, as the encoder forward uses
static __device__ __align__(256) double scratch[(size_t)kStaticMaxBatch * kStaticNsplit * 2];   // 1 MiB per GPU

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static double* g_scratch_ptr[kMaxDevices];      // per device: where that device's copy of `scratch` lives

// The array's name is only a host-side stand-in; its device address is looked up once per device, with that
// device current, and only that pointer is ever handed to a kernel.
static double* scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    fail("static scratch: no current device");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      fail("static scratch: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = static_cast<double*>(ptr);
  }
  return g_scratch_ptr[dev];
}

int launch_lognorm_conv1_static(int dtype, const float* mel, const float* w9, const float* bias, void* out, int B, int T,
                                int F, int C, hipStream_t stream) {
  // every check comes before the first kernel is enqueued: a rejected call leaves the stream untouched
  if (B <= 0 || T <= 0 || F <= 0) return fail("lognorm_conv1_static: empty input");
  if (B > kStaticMaxBatch)
    return fail("lognorm_conv1_static: %d utterances exceed the static scratch (at most %d per call)", B, kStaticMaxBatch);
  if (!mel || !w9 || !bias || !out) return fail("lognorm_conv1_static: null operand");
  if (dtype != DT_BF16 && dtype != DT_F32) return fail("lognorm_conv1_static: dtype %d", dtype);
  if (((long long)T * F) % 4 || (long long)T * F >= (1ll << 31)) return fail("lognorm_conv1_static: T*F=%lld must be a multiple of 4 below 2^31", (long long)T * F);
  if (C % 8 || 256 % (C / 8) || C > 2048) return fail("lognorm_conv1_static: C=%d must be 8 * a divisor of 256", C);
  // the scratch address belongs to the CURRENT device; a stream of another GPU would hand that GPU's kernels an
  // address that means something else there
  int cur = -1;
  hipDevice_t sdev = -1;
  if (hipGetDevice(&cur) != hipSuccess) return fail("lognorm_conv1_static: no current device");
  if (hipStreamGetDevice(stream, &sdev) != hipSuccess) return fail("lognorm_conv1_static: cannot query the stream's device");
  if ((int)sdev != cur)
    return fail("lognorm_conv1_static: the stream belongs to device %d but device %d is current", (int)sdev, cur);
  double* partial = scratch_for_current_device();
  if (!partial) return -1;
  // nothing in the array has to start at zero: the statistics kernel writes every {sum, sum of squares} pair
  // that conv1 reads (B * kStaticNsplit of them) before conv1 runs on the same stream
  if (launch_input_stats(mel, partial, B, T, F, kStaticNsplit, stream)) return -1;
  return launch_conv1(dtype, mel, partial, kStaticNsplit, w9, bias, out, B, T, F, C, stream);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension (eps 1e-5, affine), one wave per token, fp32 statistics
// (mean, then centred variance, as nn.LayerNorm).  Optionally chains a second LayerNorm on the
// first one's output: the block's closing LayerNorm (model.py:677) followed by the next block's
// ff1.layer_norm (model.py:578) in one pass over the residual stream.
// ---------------------------------------------------------------------------------------------
template <typename T, int NV>   // NV float4 per lane, d <= 256*NV
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, const float* __restrict__ g1,
                                                        const float* __restrict__ b1, const float* __restrict__ g2,
                                                        const float* __restrict__ b2, float* __restrict__ Xout,
                                                        T* __restrict__ Aout, float* __restrict__ compact, int M,
                                                        int d, int Tp, int Tvalid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float inv_d = 1.0f / (float)d;
  f32x4 v[NV];
  bool on[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    on[i] = c < d;
    v[i] = on[i] ? *reinterpret_cast<const f32x4*>(X + (size_t)row * d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float mean = s * inv_d, q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (on[i])
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float c = v[i][k] - mean;
        q += c * c;
      }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  float rstd = rsqrtf(q * inv_d + 1e-5f);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (on[i]) {
      const int c = i * 256 + lane * 4;
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g1 + c), bb = *reinterpret_cast<const f32x4*>(b1 + c);
      v[i] = (v[i] - mean) * rstd * gg + bb;
    }
  if (Xout) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i]) *reinterpret_cast<f32x4*>(Xout + (size_t)row * d + i * 256 + lane * 4) = v[i];
  }
  if (compact) {
    const int b = row / Tp, t = row - b * Tp;
    if (t < Tvalid) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (on[i])
          *reinterpret_cast<f32x4*>(compact + ((size_t)b * Tvalid + t) * d + i * 256 + lane * 4) = v[i];
    }
  }
  if (!Aout) return;
  if (g2) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i]) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    mean = s * inv_d;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i])
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float c = v[i][k] - mean;
          q += c * c;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    rstd = rsqrtf(q * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i]) {
        const int c = i * 256 + lane * 4;
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g2 + c), bb = *reinterpret_cast<const f32x4*>(b2 + c);
        v[i] = (v[i] - mean) * rstd * gg + bb;
      }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (on[i]) store4(Aout + (size_t)row * d + i * 256 + lane * 4, v[i]);
}

template <typename T>
static int launch_ln_t(const LayerNormArgs& a, hipStream_t stream) {
  const dim3 grid((a.M + 3) / 4), block(256);
  const int nv = (a.d + 255) / 256;
#define FCENC_LN(NV)                                                                                             \
  hipLaunchKernelGGL((layernorm_kernel<T, NV>), grid, block, 0, stream, a.X, a.g1, a.b1, a.g2, a.b2, a.Xout,      \
                     (T*)a.Aout, a.compact, a.M, a.d, a.Tp, a.Tvalid)
  switch (nv) {
    case 1: FCENC_LN(1); break;
    case 2: FCENC_LN(2); break;
    case 3: FCENC_LN(3); break;
    case 4: FCENC_LN(4); break;
    default: return fail("layernorm: d=%d too large", a.d);
  }
#undef FCENC_LN
  return check_launch("layernorm");
}

int launch_layernorm(int dtype, const LayerNormArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.d <= 0 || a.d % 4) return fail("layernorm: bad shape M=%d d=%d", a.M, a.d);
  if (!a.X || !a.g1 || !a.b1) return fail("layernorm: null operand");
  if ((a.g2 == nullptr) != (a.b2 == nullptr)) return fail("layernorm: second affine incomplete");
  if (a.compact && (a.Tp <= 0 || a.Tvalid <= 0 || a.Tvalid > a.Tp || a.M % a.Tp)) return fail("layernorm: compact geometry");
  if (dtype == DT_BF16) return launch_ln_t<bf16>(a, stream);
  if (dtype == DT_F32) return launch_ln_t<float>(a, stream);
  return fail("layernorm: dtype %d", dtype);
}

// ---------------------------------------------------------------------------------------------
// causal depthwise conv, kernel 9, with BatchNorm1d(eval) folded into (w9, bias), then SiLU:
//   y[b,t,c] = silu(bias[c] + sum_m w9[m][c] * x[b, t-8+m, c]),  x = 0 for t-8+m < 0
// (model.py:236-249 zero left pad, :523-538).  Layout [B][Tp][d]; a thread owns 8 channels and
// a run of DW_RUN frames, sliding a 9-row register window down time.
// ---------------------------------------------------------------------------------------------
constexpr int DW_RUN = 4;   // frames per thread: short runs keep >= 8 waves per CU at B*Tp = 16k frames

template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ in, const float* __restrict__ w9,
                                                     const float* __restrict__ bias, T* __restrict__ out, int Tp,
                                                     int d) {
  constexpr int RUN = DW_RUN;
  const int groups = d >> 3;
  const int cg = threadIdx.x % groups, tl = threadIdx.x / groups, ntl = blockDim.x / groups;
  const int b = blockIdx.y;
  const int t0 = (blockIdx.x * ntl + tl) * RUN;
  if (t0 >= Tp) return;
  const int c0 = cg * 8;
  float w[9][8], bb[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) load8<float>(w9 + (size_t)k * d + c0, w[k]);
  load8<float>(bias + c0, bb);
  const T* base = in + (size_t)b * Tp * d + c0;
  T* obase = out + (size_t)b * Tp * d + c0;
  float win[9][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int t = t0 - 8 + k;
    if (t >= 0) load8<T>(base + (size_t)t * d, win[k]);
    else
#pragma unroll
      for (int q = 0; q < 8; ++q) win[k][q] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < RUN; ++s) {
    const int t = t0 + s;
    if (t >= Tp) break;
    load8<T>(base + (size_t)t * d, win[(8 + s) % 9]);
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = bb[q];
#pragma unroll
    for (int m = 0; m < 9; ++m)
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = fmaf(w[m][q], win[(s + m) % 9][q], y[q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = silu_f(y[q]);
    store4(obase + (size_t)t * d, f32x4{y[0], y[1], y[2], y[3]});
    store4(obase + (size_t)t * d + 4, f32x4{y[4], y[5], y[6], y[7]});
  }
}

int launch_dwconv(int dtype, const void* in, const float* w9, const float* bias, void* out, int B, int Tp, int d,
                  hipStream_t stream) {
  if (B <= 0 || Tp <= 0 || d <= 0 || d % 8) return fail("dwconv: bad shape");
  const int groups = d / 8;
  if (groups > 256) return fail("dwconv: d=%d too large", d);
  const int ntl = 256 / groups;
  const dim3 block(groups * ntl), grid((Tp + DW_RUN * ntl - 1) / (DW_RUN * ntl), B);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(dwconv_kernel<bf16>, grid, block, 0, stream, (const bf16*)in, w9, bias, (bf16*)out, Tp, d);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(dwconv_kernel<float>, grid, block, 0, stream, (const float*)in, w9, bias, (float*)out, Tp, d);
  else
    return fail("dwconv: dtype %d", dtype);
  return check_launch("dwconv");
}

// ---------------------------------------------------------------------------------------------
// lengths: out_len = three rounds of ceil(n/2) (model.py:755-760), mask[b][t] = t >= out_len[b]
// (model.py:872-876, True = padding).  Integer work, bit-exact by construction.
// ---------------------------------------------------------------------------------------------
// The same launch can clear a byte range per utterance (the layout-padding frames of the last
// subsampling activation, which must be finite before they are projected): zero_bytes bytes at
// zero_base + b * zero_pitch, all multiples of 16.
__global__ void lengths_kernel(const int64_t* __restrict__ lengths, int64_t* __restrict__ out_len,
                               uint8_t* __restrict__ mask, int32_t* __restrict__ klen, int B, int Tvalid,
                               char* __restrict__ zero_base, size_t zero_pitch, size_t zero_bytes) {
  const int b = blockIdx.x;
  if (zero_base) {
    u32x4* z = reinterpret_cast<u32x4*>(zero_base + (size_t)b * zero_pitch);
    for (size_t i = threadIdx.x; i < zero_bytes / 16; i += blockDim.x) z[i] = u32x4{0u, 0u, 0u, 0u};
  }
  int64_t n = Tvalid;
  if (lengths) {
    n = lengths[b];
    for (int i = 0; i < 3; ++i) n = (n + 1) >> 1;   // floor((n+1)/2), also for n <= 0 as Python's //
  }
  if (threadIdx.x == 0) {
    out_len[b] = n;
    int64_t k = lengths ? n : Tvalid;
    k = k < 0 ? 0 : (k > Tvalid ? Tvalid : k);
    klen[b] = (int32_t)k;
  }
  if (mask)
    for (int t = threadIdx.x; t < Tvalid; t += blockDim.x) mask[(size_t)b * Tvalid + t] = (int64_t)t >= n ? 1 : 0;
}

int launch_lengths(const int64_t* lengths, int64_t* out_len, uint8_t* mask, int32_t* klen, int B, int Tvalid,
                   hipStream_t stream, void* zero_base, size_t zero_pitch, size_t zero_bytes) {
  if (B <= 0 || Tvalid <= 0 || !out_len || !klen) return fail("lengths: bad arguments");
  if (zero_base && (((uintptr_t)zero_base | zero_pitch | zero_bytes) & 15)) return fail("lengths: unaligned zero range");
  hipLaunchKernelGGL(lengths_kernel, dim3(B), dim3(256), 0, stream, lengths, out_len, mask, klen, B, Tvalid,
                     (char*)zero_base, zero_pitch, zero_bytes);
  return check_launch("lengths");
}

}  // namespace fcenc
