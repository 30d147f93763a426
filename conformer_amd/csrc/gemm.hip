// NT GEMM  out[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogues) and the 3x3 stride-2
// NHWC convolution as an implicit GEMM, for gfx950.
//
// Tile: 128 tokens x 128 features per 256-thread workgroup, K cut into 128-byte slices
// (64 bf16 / 32 fp32), two LDS stages (64 KiB) with the next slice held in registers while the
// current one is consumed (issue-early / write-late).  The weights are the MFMA "A" operand
// and the activations the "B" operand, so a lane ends up with FOUR CONSECUTIVE output features
// of one token in each accumulator and the epilogue stores 8 (bf16) or 16 (fp32) bytes per lane.
//
// Wave layout: 4 waves as 2 (features) x 2 (tokens), each 64 x 64 = 4 x 4 accumulators.
// Reference semantics: nn.Linear / 1x1 Conv1d (model.py:348-350, 429, 520, 541, 579-582, 751)
// and Conv2d(3x3, s2, p1)+BatchNorm2d(eval)+SiLU (model.py:709-719).
#include "common.h"
#include "kernels.h"

namespace fcenc {

struct GemmDev {
  const char* A;
  const char* W;
  const float* bias;
  char* out;
  float* X;
  char* Xc;
  int M, N, K, lda, ldo;
  float alpha, beta;
  // convolution geometry (CONV only)
  int Hin, Win, Cin, Hout, Wout, HoutPad;
};

template <typename T, int EPI, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmDev p) {
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;   // elements per 128-byte slice row
  constexpr int KS = BK / 32;   // k32 steps per slice
  using Frag = typename Traits<T>::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int wn = wave >> 1, wm = wave & 1;

  const int ntn = p.N >> 7;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (tile % ntn) << 7;
  const int m0 = (tile / ntn) << 7;

  // ---- per-thread staging coordinates: 4 rows (rr + 32 i) x one 16-byte chunk (cc) per tile ----
  const int cc = tid & 7, rr = tid >> 3;
  const char* wptr[4];
  const char* aptr[4];
  bool avalid[4];
  int hb[4], wb[4], pix[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = rr + 32 * i;
    wptr[i] = p.W + ((size_t)(n0 + row) * p.K + cc * EPC) * sizeof(T);
    const int m = m0 + row;
    avalid[i] = m < p.M;
    if constexpr (CONV) {
      const int mm = avalid[i] ? m : 0;
      const int wo = mm % p.Wout, t = mm / p.Wout;
      const int ho = t % p.Hout, b = t / p.Hout;
      hb[i] = 2 * ho - 1;
      wb[i] = 2 * wo - 1;
      pix[i] = b * p.Hin * p.Win;
      aptr[i] = nullptr;
    } else {
      aptr[i] = p.A + ((size_t)(avalid[i] ? m : 0) * p.lda + cc * EPC) * sizeof(T);
      hb[i] = wb[i] = pix[i] = 0;
    }
  }
  const int tiles_per_tap = CONV ? p.Cin / BK : 1;

  u32x4 wreg[4], areg[4];
  auto load_slice = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wreg[i] = *reinterpret_cast<const u32x4*>(wptr[i] + (size_t)kt * BK * sizeof(T));
    if constexpr (CONV) {
      const int tap = kt / tiles_per_tap, c0 = (kt - tap * tiles_per_tap) * BK;
      const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hi = hb[i] + kh, wi = wb[i] + kw;
        const bool ok = avalid[i] && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok)
          v = *reinterpret_cast<const u32x4*>(
              p.A + ((size_t)(pix[i] + hi * p.Win + wi) * p.Cin + c0 + cc * EPC) * sizeof(T));
        areg[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (avalid[i]) v = *reinterpret_cast<const u32x4*>(aptr[i] + (size_t)kt * BK * sizeof(T));
        areg[i] = v;
      }
    }
  };
  auto store_slice = [&](int stage) {
    char* wt = smem + stage * 32768;
    char* at = wt + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t off = tile_off<128>(rr + 32 * i, cc);
      *reinterpret_cast<u32x4*>(wt + off) = wreg[i];
      *reinterpret_cast<u32x4*>(at + off) = areg[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  load_slice(0);
  store_slice(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_slice(kt + 1);
    const char* wt = smem + (kt & 1) * 32768;
    const char* at = wt + 16384;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Frag wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = lds_frag<T, 128>(wt, wn * 64 + i * 16 + r, ks, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = lds_frag<T, 128>(at, wm * 64 + j * 16 + r, ks, g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc[i][j], wf[i], af[j]);
    }
    if (more) store_slice((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds features n..n+3 of token m for every (i, j) ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + r;
    if (m >= p.M) continue;
    size_t orow = (size_t)m;
    if constexpr (CONV) {
      const int wo = m % p.Wout, t = m / p.Wout;
      const int ho = t % p.Hout, b = t / p.Hout;
      orow = ((size_t)b * p.HoutPad + ho) * p.Wout + wo;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + 4 * g;
      f32x4 v = acc[i][j];
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if constexpr (EPI == EPI_STORE) {
        store4(reinterpret_cast<T*>(p.out) + orow * p.ldo + n, v);
      } else if constexpr (EPI == EPI_STORE_SILU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
        store4(reinterpret_cast<T*>(p.out) + orow * p.ldo + n, v);
      } else if constexpr (EPI == EPI_GLU) {
        if ((i & 1) == 0) {
          f32x4 gate = acc[(i + 1) & 3][j];
          if (p.bias) gate += *reinterpret_cast<const f32x4*>(p.bias + n + 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v[q] * sigmoid_f(gate[q]);
          const int c = (n0 >> 1) + wn * 32 + (i >> 1) * 16 + 4 * g;
          store4(reinterpret_cast<T*>(p.out) + orow * p.ldo + c, v);
        }
      } else {  // EPI_RESID
        float* xp = p.X + orow * p.N + n;
        f32x4 x = v * p.alpha;
        if (p.beta != 0.f) x += p.beta * *reinterpret_cast<const f32x4*>(xp);
        *reinterpret_cast<f32x4*>(xp) = x;
        if (p.Xc) store4(reinterpret_cast<T*>(p.Xc) + orow * p.N + n, x);
      }
    }
  }
}

template <typename T, int EPI, bool CONV>
static int launch_t(const GemmDev& d, hipStream_t stream) {
  auto k = gemm_kernel<T, EPI, CONV>;
  const int grid = ((d.M + 127) / 128) * (d.N / 128);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, stream, d);
  return check_launch("gemm");
}

template <typename T>
static int launch_gemm_t(int epi, const GemmDev& d, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_t<T, EPI_STORE, false>(d, s);
    case EPI_STORE_SILU: return launch_t<T, EPI_STORE_SILU, false>(d, s);
    case EPI_GLU: return launch_t<T, EPI_GLU, false>(d, s);
    case EPI_RESID: return launch_t<T, EPI_RESID, false>(d, s);
  }
  return fail("gemm: unknown epilogue %d", epi);
}

int launch_gemm(int dtype, int epi, const GemmArgs& a, hipStream_t stream) {
  const int bk = dtype == DT_BF16 ? 64 : 32;
  if (dtype != DT_BF16 && dtype != DT_F32) return fail("gemm: dtype %d", dtype);
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail("gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  if (a.N % 128 || a.K % bk) return fail("gemm: N=%d must be a multiple of 128 and K=%d of %d", a.N, a.K, bk);
  if (a.lda < a.K || (a.lda * (dtype == DT_BF16 ? 2 : 4)) % 16) return fail("gemm: lda=%d", a.lda);
  if (!a.A || !a.W) return fail("gemm: null operand");
  GemmDev d{};
  d.A = (const char*)a.A; d.W = (const char*)a.W; d.bias = a.bias; d.out = (char*)a.out;
  d.X = a.X; d.Xc = (char*)a.Xc; d.M = a.M; d.N = a.N; d.K = a.K; d.lda = a.lda; d.ldo = a.ldo;
  d.alpha = a.alpha; d.beta = a.beta;
  if (epi == EPI_RESID) {
    if (!a.X) return fail("gemm: residual epilogue needs X");
  } else {
    if (!a.out) return fail("gemm: null output");
    const int cols = epi == EPI_GLU ? a.N / 2 : a.N;
    if (a.ldo < cols || a.ldo % 4) return fail("gemm: ldo=%d", a.ldo);
  }
  return dtype == DT_BF16 ? launch_gemm_t<bf16>(epi, d, stream) : launch_gemm_t<float>(epi, d, stream);
}

int launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t stream) {
  const int bk = dtype == DT_BF16 ? 64 : 32;
  if (dtype != DT_BF16 && dtype != DT_F32) return fail("conv: dtype %d", dtype);
  if (a.B <= 0 || a.Hin <= 0 || a.Win <= 0) return fail("conv: empty input");
  if (a.Cin % bk || a.Cout % 128) return fail("conv: Cin=%d must be a multiple of %d, Cout=%d of 128", a.Cin, bk, a.Cout);
  if (a.Hout != (a.Hin + 1) / 2 || a.Wout != (a.Win + 1) / 2 || a.HoutPad < a.Hout)
    return fail("conv: output geometry %dx%d (pad %d) does not match input %dx%d", a.Hout, a.Wout, a.HoutPad, a.Hin, a.Win);
  const long long pixels = (long long)a.B * a.Hout * a.Wout;
  if (pixels >= (1ll << 31) || (long long)a.B * a.Hin * a.Win >= (1ll << 31)) return fail("conv: too many pixels");
  GemmDev d{};
  d.A = (const char*)a.in; d.W = (const char*)a.W; d.bias = a.bias; d.out = (char*)a.out;
  d.M = (int)pixels; d.N = a.Cout; d.K = 9 * a.Cin; d.lda = 0; d.ldo = a.Cout;
  d.Hin = a.Hin; d.Win = a.Win; d.Cin = a.Cin; d.Hout = a.Hout; d.Wout = a.Wout; d.HoutPad = a.HoutPad;
  return dtype == DT_BF16 ? launch_t<bf16, EPI_STORE_SILU, true>(d, stream)
                          : launch_t<float, EPI_STORE_SILU, true>(d, stream);
}

}  // namespace fcenc
