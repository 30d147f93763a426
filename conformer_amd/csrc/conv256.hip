// 3x3 stride-2 pad-1 NHWC convolution + bias + SiLU as an implicit GEMM with a 256 x 256 tile
// (bf16 MFMA, gfx950): the subsampling convs conv2 / conv3 (model.py:709-719), Cout = 256.
//
// Why a second tile shape: with the generic 128 x 128 kernel the im2col operand is fetched
// 2 (feature tiles) x 9/4 (taps sharing an input pixel) times from beyond L2 and the kernel
// runs at the memory hierarchy's pace, not the MFMA's.  Covering all 256 output channels and 256
// pixels per workgroup halves the operand fetches per FLOP twice over.
//
//   workgroup: 512 threads = 8 waves as 2 (pixels) x 4 (channels); wave tile 128 pixels x 64
//              channels = 4 x 8 accumulators (128 VGPRs)
//   LDS:       2 W tiles + 3 im2col tiles of 256 x 64 bf16 = 160 KiB, filled by direct-to-LDS loads
//              (source-side XOR swizzle); the W slice is requested one slice ahead (it always hits
//              L2), the im2col slice two slices ahead (it comes from HBM)
//   K loop:    36 slices of 64, channel chunk outer / tap inner so that the taps' overlapping
//              re-reads stay close enough in time for L2 to serve them (measured: fabric reads
//              halve against tap-outer order); per slice 4 steps (k32 half x pixel half) of 16
//              MFMAs, fragments read by inline asm one step ahead and awaited with counted lgkmcnt
//              so LDS latency hides behind the MFMAs
//   epilogue:  two passes (one per pixel half) through a 128 x 256 fp32 LDS tile so that the
//              bf16 output rows are written 16 bytes per lane, 512 contiguous bytes per row
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int TILE = 32768;                 // one 256-row x 64-column bf16 operand tile
constexpr int W_STAGES = 2, A_STAGES = 3;   // weights always hit L2; im2col rows come from HBM and get one more slice of lead
constexpr int A_BASE = W_STAGES * TILE;
constexpr int LDS_BYTES = (W_STAGES + A_STAGES) * TILE;

__device__ __attribute__((aligned(16))) unsigned int zero16[4];   // source of padding taps / rows beyond M

// FCENC_C2_NOLOAD (_A / _W: im2col or weight tiles only) / FCENC_C2_NOLDSREAD: timing experiments (wrong results) that drop the direct-to-LDS loads / the
// fragment reads, to see what each costs on top of the MFMAs and barriers (tools/diag_build.sh c2_*).
__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
#ifndef FCENC_C2_NOLOAD
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

template <int IMM>
__device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
#ifdef FCENC_C2_NOLDSREAD
  asm volatile("" : "=v"(dst) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
#endif
}

struct ConvDev {
  const char* in;
  const char* W;
  const float* bias;
  bf16* out;
  int M, Cin, Cout, Hin, Win, Hout, Wout, HoutPad;
};

__device__ __forceinline__ float* ctile(float* ct, int row, int col) {   // [128][256] fp32, chunk ^ (row & 7)
  return ct + row * 256 + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3));
}

__global__ __launch_bounds__(512) void conv256_kernel(const ConvDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;

  const int ntn = p.Cout >> 8;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (tile % ntn) << 8;
  const int m0 = (tile / ntn) << 8;
  const int K = 9 * p.Cin;
  const int nk = K >> 6;
  const int tiles_per_tap = p.Cin >> 6;

  // ---- loader: wave w stages rows 32w..32w+31 of both tiles (4 + 4 wave-instructions) ----
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const char* wsrc = p.W + (size_t)(n0 + 32 * wave) * K * 2;
  const uint32_t woff = (uint32_t)((srow * K + schunk * 8) * 2);
  const size_t wstep = (size_t)8 * K * 2;
  // im2col coordinates of the 4 pixels this lane stages: byte offset of the pixel's tap (0,0)
  // relative to the image of the tile's first pixel (may be negative for border pixels; adding an
  // in-image tap offset makes it valid) and a 9-bit mask of the taps that fall inside the image.
  const int hw_out = p.Hout * p.Wout;
#ifdef FCENC_C2_SAMEA
  const int ma = 0;       // timing experiment (wrong results): every workgroup stages the first tile's im2col rows (all L2 hits)
#else
  const int ma = m0;
#endif
  const int b0 = ma / hw_out;
  const char* inbase = p.in + (size_t)b0 * p.Hin * p.Win * p.Cin * 2;
  int aoff[4];
  uint32_t amask[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = ma + 32 * wave + 8 * q + srow;
    const int mm = m < p.M ? m : ma;
    const int wo = mm % p.Wout, t = mm / p.Wout;
    const int ho = t % p.Hout, b = t / p.Hout;
    const int hq = 2 * ho - 1, wq = 2 * wo - 1;
    aoff[q] = (((b - b0) * p.Hin + hq) * p.Win + wq) * p.Cin * 2 + schunk * 16;
    uint32_t mk = 0;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int hi = hq + tp / 3, wi = wq + tp % 3;
      if (m < p.M && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win) mk |= 1u << tp;
    }
    amask[q] = mk;
  }
  const char* zsrc = reinterpret_cast<const char*>(zero16);
  // K is walked channel-chunk outer, tap inner: the nine taps of a chunk re-read overlapping pixels,
  // and keeping those re-reads a few slices apart lets the XCD's L2 serve them (with the tap outer
  // the re-read distance is four times longer and every tap goes back to HBM).
  // `half` selects rows 0-15 or 16-31 of the wave's 32-row share (two wave-instructions)
  auto issue_a = [&](int kt_raw, int half) {
    const int kt = kt_raw < nk ? kt_raw : nk - 1;
    const int cc = kt / 9, tap = kt - 9 * cc;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int tapoff = ((kh * p.Win + kw) * p.Cin + (cc << 6)) * 2;      // wave-uniform
    char* at = smem + A_BASE + (kt_raw % A_STAGES) * TILE + wave * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((q >> 1) != half) continue;
      const bool ok = (amask[q] >> tap) & 1u;
#ifndef FCENC_C2_NOLOAD_A
      glds16(ok ? inbase + (uint32_t)(aoff[q] + tapoff) : zsrc, at + q * 1024);
#endif
    }
  };
  auto issue_w = [&](int kt_raw, int half) {
    const int kt = kt_raw < nk ? kt_raw : nk - 1;
    const int cc = kt / 9, tap = kt - 9 * cc;
    const size_t kcol = (size_t)(tap * tiles_per_tap + cc) * 128;
    char* wt = smem + (kt_raw & 1) * TILE + wave * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
#ifndef FCENC_C2_NOLOAD_W
      if ((q >> 1) == half) glds16(wsrc + q * wstep + kcol + woff, wt + q * 1024);
#endif
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // gate(kt): slice kt landed everywhere (only the four im2col loads of slice kt+1, the youngest, may
  // still be in flight) and every wave is done reading slice kt-1, whose tiles may now be refilled.
  // The refill (W of kt+1, then im2col of kt+2) is spread over the next four MFMA steps, two
  // wave-instructions each, so that a wave's load issue overlaps its SIMD partner's MFMAs.
  auto gate = [&]() {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t wbase[2], abase[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t swz = (uint32_t)((((ks << 2) | g) ^ (r & 7)) << 4);
    wbase[ks] = lds0 + (uint32_t)(wn * 64 + r) * 128u + swz;
    abase[ks] = lds0 + (uint32_t)A_BASE + (uint32_t)(wm * 128 + r) * 128u + swz;
  }
  u32x4 wf[2][4], af[2][4];        // W fragments by k32 half, pixel fragments by step parity
  auto load_w = [&](int kt, auto ksc) {
    constexpr int ks = decltype(ksc)::value;
    const uint32_t a = wbase[ks] + (uint32_t)(kt & 1) * TILE;
    lds_read128<0>(wf[ks][0], a);
    lds_read128<2048>(wf[ks][1], a);
    lds_read128<4096>(wf[ks][2], a);
    lds_read128<6144>(wf[ks][3], a);
  };
  auto load_a = [&](uint32_t stage_off, auto ksc, auto hc) {
    constexpr int ks = decltype(ksc)::value, h = decltype(hc)::value;
    const uint32_t a = abase[ks] + stage_off;
    lds_read128<h * 8192 + 0>(af[h][0], a);
    lds_read128<h * 8192 + 2048>(af[h][1], a);
    lds_read128<h * 8192 + 4096>(af[h][2], a);
    lds_read128<h * 8192 + 6144>(af[h][3], a);
  };
  // MFMAs of step (ks, h); `pending` = fragment reads issued after this step's own (0, 4 or 8)
  auto mma_step = [&](auto ksc, auto hc, auto pendc, auto&& refill) {
    constexpr int ks = decltype(ksc)::value, h = decltype(hc)::value, pending = decltype(pendc)::value;
    static_assert(pending == 0 || pending == 4 || pending == 8, "fragment reads come in fours");
#define FCENC_FRAGS "+v"(wf[ks][0]), "+v"(wf[ks][1]), "+v"(wf[ks][2]), "+v"(wf[ks][3]), "+v"(af[h][0]), "+v"(af[h][1]), "+v"(af[h][2]), "+v"(af[h][3])
    if constexpr (pending == 8) asm volatile("s_waitcnt lgkmcnt(8)" : FCENC_FRAGS);
    else if constexpr (pending == 4) asm volatile("s_waitcnt lgkmcnt(4)" : FCENC_FRAGS);
    else asm volatile("s_waitcnt lgkmcnt(0)" : FCENC_FRAGS);
#undef FCENC_FRAGS
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mma(acc[i][4 * h + j], __builtin_bit_cast(bf16x8, wf[ks][i]), __builtin_bit_cast(bf16x8, af[h][j]));
    __builtin_amdgcn_sched_barrier(0);
    refill();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        mma(acc[i][4 * h + j], __builtin_bit_cast(bf16x8, wf[ks][i]), __builtin_bit_cast(bf16x8, af[h][j]));
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;

  issue_a(0, 0); issue_a(0, 1);
  issue_w(0, 0); issue_w(0, 1);
  issue_a(1, 0); issue_a(1, 1);
  gate();
  issue_w(1, 0);
  uint32_t cur = 0;                // byte offset of the im2col stage holding slice kt
  load_w(0, I0{});
  load_a(cur, I0{}, I0{});
  // The last slice is peeled off the loop so that every wait statement sits on exactly one path: where
  // two paths with different waits merge in front of shared MFMAs, the compiler may copy the fragment
  // registers into the merged ones BEFORE the wait, i.e. before the LDS data has arrived.
  for (int kt = 0; kt + 1 < nk; ++kt) {
    load_a(cur, I0{}, I1{});
    mma_step(I0{}, I0{}, I4{}, [&] { issue_w(kt + 1, 1); });
    load_w(kt, I1{});
    load_a(cur, I1{}, I0{});
    mma_step(I0{}, I1{}, I8{}, [&] { issue_a(kt + 2, 0); });
    load_a(cur, I1{}, I1{});
    mma_step(I1{}, I0{}, I4{}, [&] { issue_a(kt + 2, 1); });
    gate();
    cur = cur == (A_STAGES - 1) * TILE ? 0u : cur + TILE;
    load_w(kt + 1, I0{});
    load_a(cur, I0{}, I0{});
    mma_step(I1{}, I1{}, I8{}, [&] { issue_w(kt + 2, 0); });
  }
  {
    const int kt = nk - 1;
    load_a(cur, I0{}, I1{});
    mma_step(I0{}, I0{}, I4{}, [&] { issue_w(kt + 1, 1); });
    load_w(kt, I1{});
    load_a(cur, I1{}, I0{});
    mma_step(I0{}, I1{}, I8{}, [&] { issue_a(kt + 2, 0); });
    load_a(cur, I1{}, I1{});
    mma_step(I1{}, I0{}, I4{}, [&] { issue_a(kt + 2, 1); });
    mma_step(I1{}, I1{}, I0{}, [] {});
  }

  // ---- epilogue: one pass per pixel half through a [128][256] fp32 LDS tile ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    __syncthreads();
    if (wm == ps) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<f32x4*>(ctile(ct, 16 * j + r, wn * 64 + 16 * i + 4 * g)) = acc[i][j];
    }
    __syncthreads();
    // not unrolled: with the eight iterations' addresses formed up front beside the 128 accumulator registers that the
    // other pixel half still holds, the kernel spilled 12 registers, and a kernel with any scratch pays for its set-up
#pragma nounroll
    for (int it = 0; it < 8; ++it) {
      const int c = tid + 512 * it;
      const int row = c >> 5, col = (c & 31) << 3;
      const int m = m0 + 128 * ps + row;
      if (m >= p.M) continue;
      const int wo = m % p.Wout, t = m / p.Wout;
      const int ho = t % p.Hout, b = t / p.Hout;
      const size_t orow = ((size_t)b * p.HoutPad + ho) * p.Wout + wo;
      const int n = n0 + col;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col)) + *reinterpret_cast<const f32x4*>(p.bias + n);
      f32x4 v1 = *reinterpret_cast<const f32x4*>(ctile(ct, row, col + 4)) + *reinterpret_cast<const f32x4*>(p.bias + n + 4);
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q] = (bf16)silu_f(v0[q]);
        o[4 + q] = (bf16)silu_f(v1[q]);
      }
      *reinterpret_cast<bf16x8*>(p.out + orow * p.Cout + n) = o;
    }
  }
}

}  // namespace

bool conv256_supported(int dtype, const ConvArgs& a) {
  return dtype == DT_BF16 && a.Cout % 256 == 0 && a.Cin % 64 == 0 && a.bias != nullptr;
}

int launch_conv256(const ConvArgs& a, hipStream_t stream) {
  const long long pixels = (long long)a.B * a.Hout * a.Wout;
  ConvDev d{};
  d.in = (const char*)a.in; d.W = (const char*)a.W; d.bias = a.bias; d.out = (bf16*)a.out;
  d.M = (int)pixels; d.Cin = a.Cin; d.Cout = a.Cout; d.Hin = a.Hin; d.Win = a.Win; d.Hout = a.Hout; d.Wout = a.Wout;
  d.HoutPad = a.HoutPad;
  if (reserve_lds(reinterpret_cast<const void*>(conv256_kernel), LDS_BYTES) != hipSuccess)
    return fail("conv256: cannot reserve %d bytes of LDS", LDS_BYTES);
  const int grid = (int)((pixels + 255) / 256) * (a.Cout / 256);
  hipLaunchKernelGGL(conv256_kernel, dim3(grid), dim3(512), LDS_BYTES, stream, d);
  return check_launch("conv256");
}

}  // namespace fcenc
