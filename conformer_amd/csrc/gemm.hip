// NT GEMM  out[m][n] = sum_k A[m][k] * W[n][k]  (+ fused epilogues) and the 3x3 stride-2
// NHWC convolution as an implicit GEMM, for gfx950.
//
// Tile: 128 tokens x 128 features per workgroup of NW waves (8: 2 x 4 waves of 64 features x 32
// tokens; 4: 2 x 2 of 64 x 64), K cut into 128-byte slices (64 bf16 / 32 fp32).
//
// Staging: a ring of NST LDS stages (32 KiB each) filled by direct-to-LDS loads
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers).  NST-1 slices are
// always in flight: each iteration waits with a COUNTED vmcnt for the oldest slice only, passes
// one raw s_barrier (which also proves that every wave has finished reading the stage about to
// be refilled), issues the slice NST-1 steps ahead and then computes.  The LDS image is linear
// per wave-instruction, so the bank swizzle (chunk ^ (row & 7)) is applied to the per-lane SOURCE
// address and undone by the same XOR in the fragment reads.  Past the end of K the loader
// re-issues the last slice into the free stage so that the in-flight count stays constant.
//
// The weights are the MFMA "A" operand and the activations the "B" operand, so a lane ends up
// with four consecutive output features of one token per accumulator.  The epilogue transposes
// the 128 x 128 fp32 tile through LDS so that every global access of the epilogue (bias, the
// fp32 residual stream, the output) is 16 bytes per lane along a row.
//
// Reference semantics: nn.Linear / 1x1 Conv1d (model.py:348-350, 429, 520, 541, 579-582, 751)
// and Conv2d(3x3, s2, p1)+BatchNorm2d(eval)+SiLU (model.py:709-719).
#include <type_traits>

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
  int splits;   // grid.y: slices of the contraction (1 = whole)
  int ldw;   // elements between rows of W (K unless the contraction is cut into slices, below)
  float alpha, beta;
  // convolution geometry (CONV only)
  int Hin, Win, Cin, Hout, Wout, HoutPad;
};

// 16 zero bytes: source of im2col taps that fall into the zero padding and of rows beyond M
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int IMM>
__device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

__device__ __forceinline__ void store8(bf16* p, const f32x4& a, const f32x4& b) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = (bf16)a[i];
    o[4 + i] = (bf16)b[i];
  }
  *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const f32x4& a, const f32x4& b) {
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

constexpr int STAGE_BYTES = 32768;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt");
}

// epilogue tile in LDS: [128 tokens][128 features] fp32, 16-byte chunk c of row m stored at c ^ (m & 7)
__device__ __forceinline__ float* ctile(float* ct, int row, int col) {
  return ct + row * 128 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3));
}

// NW waves (4: 2 x 2 of 64 x 64; 8: 2 x 4 of 64 features x 32 tokens), NST LDS stages
template <typename T, int EPI, bool CONV, int NW, int NST>
__global__ __launch_bounds__(NW * 64) void gemm_kernel(const GemmDev p) {
  constexpr int EPC = Traits<T>::EPC;
  constexpr int BK = 8 * EPC;   // elements per 128-byte slice row
  constexpr int KS = BK / 32;   // k32 steps per slice
  using Frag = typename Traits<T>::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  // scalar wave index: ring destinations and row groups stay in SGPRs, the direct-to-LDS loads
  // take a scalar base plus a per-lane 32-bit offset
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  constexpr int WM = NW / 2;          // waves along tokens
  constexpr int MF = 8 / WM;          // 16-token fragments per wave
  constexpr int RPW = 128 / NW;       // tile rows each wave stages
  constexpr int LQ = RPW / 8;         // wave-instructions per tile per wave
  const int wn = wave / WM, wm = wave % WM;

  const int ntn = p.N >> 7;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (tile % ntn) << 7;
  const int m0 = (tile / ntn) << 7;
  const int nk = p.K / BK;
  // blockIdx.y = slice of a cut contraction: this launch's K elements start at column slice * K of both operands
  // and the result goes to the slice's own [M][N] plane of X (the caller adds the planes up)
  const size_t kz_bytes = (size_t)blockIdx.y * p.K * sizeof(T);

  // ---- loader coordinates.  Wave w fills rows 32w..32w+31 of both tiles with 4 + 4
  // wave-instructions of 8 rows each; lane l writes LDS chunk (l & 7) of row (l >> 3) of its
  // group and therefore fetches source chunk (l & 7) ^ (l >> 3).
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const char* wsrc = p.W + (size_t)(n0 + RPW * wave) * p.ldw * sizeof(T) + kz_bytes;     // wave-uniform
  const uint32_t woff = (uint32_t)((srow * p.ldw + schunk * EPC) * sizeof(T));   // per lane
  const size_t wstep = (size_t)8 * p.ldw * sizeof(T);
  const char* asrc[LQ];
  int astep[LQ], hb[LQ], wb[LQ], pix[LQ];
  bool avalid[LQ];
#pragma unroll
  for (int q = 0; q < LQ; ++q) {
    const int m = m0 + RPW * wave + 8 * q + srow;
    avalid[q] = m < p.M;
    if constexpr (CONV) {
      const int mm = avalid[q] ? m : 0;
      const int wo = mm % p.Wout, t = mm / p.Wout;
      const int ho = t % p.Hout, b = t / p.Hout;
      hb[q] = 2 * ho - 1;
      wb[q] = 2 * wo - 1;
      pix[q] = b * p.Hin * p.Win;
      asrc[q] = nullptr;
      astep[q] = 0;
    } else {
      asrc[q] = avalid[q] ? p.A + ((size_t)m * p.lda + schunk * EPC) * sizeof(T) + kz_bytes : reinterpret_cast<const char*>(g_zero16);
      astep[q] = avalid[q] ? 128 : 0;
      hb[q] = wb[q] = pix[q] = 0;
    }
  }
  const int tiles_per_tap = CONV ? p.Cin / BK : 1;

  auto issue = [&](int kt_raw, int buf) {
    const int kt = kt_raw < nk ? kt_raw : nk - 1;
    char* wt = smem + buf * STAGE_BYTES + wave * (RPW * 128);
    char* at = wt + 16384;
#pragma unroll
    for (int q = 0; q < LQ; ++q) glds16(wsrc + q * wstep + (size_t)kt * 128 + woff, wt + q * 1024);
    if constexpr (CONV) {
      const int tap = kt / tiles_per_tap, c0 = (kt - tap * tiles_per_tap) * BK;
      const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int q = 0; q < LQ; ++q) {
        const int hi = hb[q] + kh, wi = wb[q] + kw;
        const bool ok = avalid[q] && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        const char* src = ok ? p.A + ((size_t)(pix[q] + hi * p.Win + wi) * p.Cin + c0 + schunk * EPC) * sizeof(T)
                             : reinterpret_cast<const char*>(g_zero16);
        glds16(src, at + q * 1024);
      }
    } else {
#pragma unroll
      for (int q = 0; q < LQ; ++q) glds16(asrc[q] + (size_t)kt * astep[q], at + q * 1024);
    }
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s, s);

  // gate(kt): slice kt has landed in every wave's part of the ring, and every wave is done reading
  // the stage that slice kt+NST-1 overwrites.  Counted vmcnt: 2*LQ loads per slice per wave, the
  // NST-2 younger slices stay in flight; lgkmcnt(0): this wave's own fragment reads are complete.
  auto gate = [&](int kt) {
    wait_vmcnt<(NST - 2) * 2 * LQ>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(kt + NST - 1, (kt + NST - 1) % NST);
  };

  if constexpr (KS == 2) {
    // bf16: software pipeline over k32 steps.  Fragment reads are inline asm with per-lane LDS
    // addresses computed once (rows 16 apart are immediates), so the reads of the NEXT step are in
    // flight while this step's MFMAs issue and a counted lgkmcnt waits only for this step's
    // fragments; the compiler on its own waits lgkmcnt(0) right after issuing any LDS read here.
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint32_t wbase[2], abase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t swz = (uint32_t)((((ks << 2) | g) ^ (r & 7)) << 4);
      wbase[ks] = lds0 + (uint32_t)(wn * 64 + r) * 128u + swz;
      abase[ks] = lds0 + 16384u + (uint32_t)(wm * (16 * MF) + r) * 128u + swz;
    }
    u32x4 wf[2][4], af[2][MF];
    auto load_step = [&](int kt, auto ksc, auto bufc) {
      constexpr int ks = decltype(ksc)::value, buf = decltype(bufc)::value;
      const uint32_t stage = (uint32_t)(kt % NST) * STAGE_BYTES;
      lds_read128<0>(wf[buf][0], wbase[ks] + stage);
      lds_read128<2048>(wf[buf][1], wbase[ks] + stage);
      lds_read128<4096>(wf[buf][2], wbase[ks] + stage);
      lds_read128<6144>(wf[buf][3], wbase[ks] + stage);
      lds_read128<0>(af[buf][0], abase[ks] + stage);
      lds_read128<2048>(af[buf][1], abase[ks] + stage);
      if constexpr (MF == 4) {
        lds_read128<4096>(af[buf][2], abase[ks] + stage);
        lds_read128<6144>(af[buf][3], abase[ks] + stage);
      }
    };
    auto mma_step = [&](auto bufc, bool more_pending) {
      constexpr int buf = decltype(bufc)::value;
      // this step's fragments are older than the 4+MF just issued for the next step
      if (more_pending) {
        if constexpr (MF == 2)
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(wf[buf][2]), "+v"(wf[buf][3]),
                       "+v"(af[buf][0]), "+v"(af[buf][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(wf[buf][2]), "+v"(wf[buf][3]),
                       "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]));
      } else {
        if constexpr (MF == 2)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(wf[buf][2]), "+v"(wf[buf][3]),
                       "+v"(af[buf][0]), "+v"(af[buf][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[buf][0]), "+v"(wf[buf][1]), "+v"(wf[buf][2]), "+v"(wf[buf][3]),
                       "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j)
          mma(acc[i][j], __builtin_bit_cast(bf16x8, wf[buf][i]), __builtin_bit_cast(bf16x8, af[buf][j]));
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // The last slice is peeled off the loop so that each wait statement sits on exactly one path: where
    // two paths with different waits merged in front of shared MFMAs, the compiler copied the fragment
    // registers into the merged ones BEFORE the wait, i.e. possibly before the LDS data had arrived.
    gate(0);
    load_step(0, I0{}, I0{});
    for (int kt = 0; kt + 1 < nk; ++kt) {
      load_step(kt, I1{}, I1{});
      mma_step(I0{}, true);
      gate(kt + 1);
      load_step(kt + 1, I0{}, I0{});
      mma_step(I1{}, true);
    }
    load_step(nk - 1, I1{}, I1{});
    mma_step(I0{}, true);
    mma_step(I1{}, false);
  } else {
    // fp32: the 64 exact-fp32 MFMAs per step dwarf the LDS latency, plain compiler-scheduled reads
    for (int kt = 0; kt < nk; ++kt) {
      gate(kt);
      const char* wt = smem + (kt % NST) * STAGE_BYTES;
      const char* at = wt + 16384;
      Frag wf[4], af[MF];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = lds_frag<T, 128>(wt, wn * 64 + i * 16 + r, 0, g);
#pragma unroll
      for (int j = 0; j < MF; ++j) af[j] = lds_frag<T, 128>(at, wm * (16 * MF) + j * 16 + r, 0, g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) mma(acc[i][j], wf[i], af[j]);
    }
  }

  // ---- epilogue: transpose the tile through LDS ([token][feature] fp32, pitch CP) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing refills must land before LDS is reused
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < MF; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(ctile(ct, wm * (16 * MF) + j * 16 + r, wn * 64 + i * 16 + 4 * g)) = acc[i][j];
  __syncthreads();

#pragma unroll
  for (int it = 0; it < 2048 / (NW * 64); ++it) {
    const int c = tid + NW * 64 * it;
    const int row = c >> 4, col = (c & 15) << 3;
    const int m = m0 + row;
    if (m >= p.M) continue;
    if constexpr (EPI == EPI_GLU) {
      if (col & 16) continue;          // gate columns are consumed by the item 16 columns to the left
    }
    size_t orow = (size_t)m;
    if constexpr (CONV) {
      const int wo = m % p.Wout, t = m / p.Wout;
      const int ho = t % p.Hout, b = t / p.Hout;
      orow = ((size_t)b * p.HoutPad + ho) * p.Wout + wo;
    }
    const int n = n0 + col;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col));
    f32x4 v1 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col + 4));
    if (p.bias) {
      v0 += *reinterpret_cast<const f32x4*>(p.bias + n);
      v1 += *reinterpret_cast<const f32x4*>(p.bias + n + 4);
    }
    if constexpr (EPI == EPI_STORE) {
      store8(reinterpret_cast<T*>(p.out) + orow * p.ldo + n, v0, v1);
    } else if constexpr (EPI == EPI_STORE_SILU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] = silu_f(v0[q]);
        v1[q] = silu_f(v1[q]);
      }
      store8(reinterpret_cast<T*>(p.out) + orow * p.ldo + n, v0, v1);
    } else if constexpr (EPI == EPI_GLU) {
      f32x4 g0 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col + 16));
      f32x4 g1 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col + 20));
      if (p.bias) {
        g0 += *reinterpret_cast<const f32x4*>(p.bias + n + 16);
        g1 += *reinterpret_cast<const f32x4*>(p.bias + n + 20);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v0[q] *= sigmoid_f(g0[q]);
        v1[q] *= sigmoid_f(g1[q]);
      }
      const int ch = (n0 >> 1) + ((col >> 5) << 4) + (col & 15);
      store8(reinterpret_cast<T*>(p.out) + orow * p.ldo + ch, v0, v1);
    } else {  // EPI_RESID
      float* xp = p.X + (size_t)blockIdx.y * p.M * p.N + orow * p.N + n;
      f32x4 x0 = v0 * p.alpha, x1 = v1 * p.alpha;
      if (p.beta != 0.f) {
        x0 += p.beta * *reinterpret_cast<const f32x4*>(xp);
        x1 += p.beta * *reinterpret_cast<const f32x4*>(xp + 4);
      }
      *reinterpret_cast<f32x4*>(xp) = x0;
      *reinterpret_cast<f32x4*>(xp + 4) = x1;
      if (p.Xc) store8(reinterpret_cast<T*>(p.Xc) + orow * p.N + n, x0, x1);
    }
  }
}

template <typename T, int EPI, bool CONV, int NW, int NST>
static int launch_v(const GemmDev& d, hipStream_t stream) {
  auto k = gemm_kernel<T, EPI, CONV, NW, NST>;
  constexpr int lds = NST * STAGE_BYTES < 65536 ? 65536 : NST * STAGE_BYTES;
  if (lds > 65536) {
    // more than 64 KiB of dynamic LDS needs the per-device function attribute; setting it on every
    // launch is a cheap host call and keeps multi-device processes correct
    if (reserve_lds(reinterpret_cast<const void*>(k), lds) != hipSuccess)
      return fail("gemm: cannot reserve %d bytes of LDS", lds);
  }
  const int grid = ((d.M + 127) / 128) * (d.N / 128);
  hipLaunchKernelGGL(k, dim3(grid, d.splits), dim3(NW * 64), lds, stream, d);
  return check_launch("gemm");
}

// Shape-based choice, measured on MI355X at M = 16384 .. 640000: eight waves (two per SIMD) beat
// four everywhere; the deep ring pays once K >= 1024 (FFN-2, subsampling Linear), while short-K
// GEMMs and the convolution prefer two stages = 64 KiB so that two workgroups share a CU.
template <typename T, int EPI, bool CONV>
static int launch_t(const GemmDev& d, hipStream_t stream) {
  if (!CONV && d.K >= 1024) return launch_v<T, EPI, CONV, 8, 4>(d, stream);
  return launch_v<T, EPI, CONV, 8, 2>(d, stream);
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
  const int es = dtype == DT_BF16 ? 2 : 4;
  if (dtype != DT_BF16 && dtype != DT_F32) return fail("gemm: dtype %d", dtype);
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail("gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  if (a.N % 128 || a.K % bk) return fail("gemm: N=%d must be a multiple of 128 and K=%d of %d", a.N, a.K, bk);
  const int splits = a.splits > 0 ? a.splits : 1, ldw = a.ldw > 0 ? a.ldw : a.K;
  if (a.lda < a.K * splits || (a.lda * es) % 16) return fail("gemm: lda=%d", a.lda);
  if (ldw < a.K * splits || (ldw * es) % 16) return fail("gemm: ldw=%d", ldw);
  if (splits > 1 && (epi != EPI_RESID || a.Xc || a.beta != 0.f || a.bias || splits > 65535))
    return fail("gemm: a cut contraction writes plain fp32 planes (residual epilogue, beta 0, no bias, no copy)");
  if (!a.A || !a.W) return fail("gemm: null operand");
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.bias) & 15) return fail("gemm: operands must be 16-byte aligned");
  GemmDev d{};
  d.A = (const char*)a.A; d.W = (const char*)a.W; d.bias = a.bias; d.out = (char*)a.out;
  d.X = a.X; d.Xc = (char*)a.Xc; d.M = a.M; d.N = a.N; d.K = a.K; d.lda = a.lda; d.ldo = a.ldo;
  d.alpha = a.alpha; d.beta = a.beta; d.ldw = ldw; d.splits = splits;
  if (epi == EPI_RESID) {
    if (!a.X) return fail("gemm: residual epilogue needs X");
    if (((uintptr_t)a.X | (uintptr_t)a.Xc) & 15) return fail("gemm: X must be 16-byte aligned");
  } else {
    if (!a.out) return fail("gemm: null output");
    const int cols = epi == EPI_GLU ? a.N / 2 : a.N;
    if (a.ldo < cols || (a.ldo * es) % 16 || ((uintptr_t)a.out & 15)) return fail("gemm: ldo=%d / output alignment", a.ldo);
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
  if (((uintptr_t)a.in | (uintptr_t)a.W | (uintptr_t)a.out | (uintptr_t)a.bias) & 15) return fail("conv: operands must be 16-byte aligned");
  if (conv256_supported(dtype, a)) return launch_conv256(a, stream);
  GemmDev d{};
  d.A = (const char*)a.in; d.W = (const char*)a.W; d.bias = a.bias; d.out = (char*)a.out;
  d.M = (int)pixels; d.N = a.Cout; d.K = 9 * a.Cin; d.lda = 0; d.ldo = a.Cout; d.ldw = d.K; d.splits = 1;
  d.Hin = a.Hin; d.Win = a.Win; d.Cin = a.Cin; d.Hout = a.Hout; d.Wout = a.Wout; d.HoutPad = a.HoutPad;
  return dtype == DT_BF16 ? launch_t<bf16, EPI_STORE_SILU, true>(d, stream)
                          : launch_t<float, EPI_STORE_SILU, true>(d, stream);
}

}  // namespace fcenc
