// The third subsampling convolution with its input patch staged once in LDS (bf16 MFMA, gfx950; 20 input columns,
// 256 channels):  Conv2d(256 -> 256, 3x3, s2, p1) + BatchNorm + SiLU        (model.py:715-719, :745)
//
// The implicit GEMM in conv256.hip gathers a pixel tile per tap, so that every input pixel crosses the CU's load path
// 2.25 times and the kernel is paced by those loads.  Here a workgroup (8 waves) owns 24 output rows x 10 output
// columns (240 pixels = 15 MFMA column groups) x all 256 output channels of one utterance; the 49 x 21 input pixels
// under them are loaded once, 32 channels at a time, into the patch layout of conv12.hip:
//
//   patch buffers: 4 planes (one per 16-byte piece = 8 channels) x 2 buffers x 1040 slots x 16 B.  An input row is 21
//                  slots: its 11 even padded columns, then its 10 odd ones, so that the stride-2 pixels of a tap are
//                  consecutive slots, tap and buffer are a compile-time byte offset of the read instruction, and the
//                  16 lanes x 16 B of a fragment read are 256 consecutive bytes of one plane: no bank conflicts.
//   loads        : while the MFMAs consume chunk c (9 taps x k32), chunk c + 1 is written into the other buffer, one
//                  group of 16 slots per wave and tap, from registers that were loaded three taps earlier (the input
//                  comes from HBM: the second convolution has just written many times the L2's size).  One barrier
//                  per chunk.
//   weights      : wave w multiplies output channels 32w .. 32w+31 for all 240 pixels; its fragments stream from L2
//                  into registers in consumption order (conv3p_pack_kernel), three taps deep: a wave's loads return
//                  in order, so a weight fragment cannot arrive before the HBM load that was issued ahead of it.
#include "common.h"
#include "frag_ring.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int C = 256;                    // input and output channels
constexpr int WIN = 20, WOUT = 10;        // input and output columns
constexpr int R = 24;                     // output rows per workgroup
constexpr int NPX = R * WOUT;             // 240 output pixels
constexpr int NJ = NPX / 16;              // 15 MFMA column groups
constexpr int PROWS = 2 * R + 1;          // 49 input rows under them
constexpr int ROW_SLOTS = 21;             // slots per input row: padded columns 0, 2, .., 20 then 1, 3, .., 19
constexpr int NSLOT = 1040;               // 49 * 21 = 1029, rounded up to whole groups of 16
constexpr int NGROUP = NSLOT / 16;        // 65 groups of 16 slots, loaded by one wave instruction each
constexpr int PLANE = NSLOT * 16;         // 16 640 B (a multiple of 256: every plane starts on bank 0)
constexpr int PBUF = 4 * PLANE;           // one 32-channel chunk of the patch (its planes are 2 * PLANE apart)
constexpr int LDS_BYTES = 2 * PBUF;       // 133 120 B
constexpr int NCHUNK = C / 32;
constexpr int NTAP = 9;
#ifndef FCENC_C3_RING
#define FCENC_C3_RING 3
#endif
constexpr int RING = FCENC_C3_RING, LEAD = RING - 1;  // pixel fragments in registers / read ahead of their MFMAs
constexpr int NFRAG = NTAP * NJ;          // 135 fragment reads per chunk
#ifndef FCENC_C3_WDEPTH
#define FCENC_C3_WDEPTH 3
#endif
constexpr int WDEPTH = FCENC_C3_WDEPTH;               // taps of weights per wave in registers: the one in use and the next two
#ifndef FCENC_C3_LDD
#define FCENC_C3_LDD 3
#endif
constexpr int LDD = FCENC_C3_LDD;                  // taps between a patch load and the LDS write of what it returned
constexpr int WSTEPS = NCHUNK * NTAP + WDEPTH;   // packed steps per wave, the last WDEPTH are zero padding
constexpr int OUT_TILE_BYTES = NPX * C * 2;
static_assert(OUT_TILE_BYTES <= LDS_BYTES, "the output tile reuses the patch buffers");
static_assert((2 * NTAP) % WDEPTH == 0 && NTAP % LDD == 0, "ring slots must be compile-time constants over a pair of chunks");
static_assert(NFRAG % RING == 0, "ring slots must be compile-time constants");
static_assert(8 * NTAP >= NGROUP, "every group has a (wave, tap) that loads it");

struct Conv3pDev {
  const bf16* in;          // [B][Hin][20][256]
  const u32x4* W;          // fragments [8 waves][WSTEPS][2][64 lanes]
  const float* bias;       // [256]
  bf16* out;               // [B][HoutPad][10][256]
  int Hin, Hout, HoutPad, tiles;
};

constexpr int tap_offset(int tap) {        // byte offset of tap (kh, kw) from tap (0, 0) inside a plane
  const int kh = tap / 3, kw = tap % 3;
  return (kh * ROW_SLOTS + (kw & 1) * 11 + (kw >> 1)) * 16;
}
constexpr int group_offset(int j) { return 256 * j + 512 * (16 * j / WOUT); }   // see pbase in the kernel
static_assert(PLANE + tap_offset(8) + group_offset(NJ - 1) < 65536, "buffer + tap + group must fit the 16-bit offset of a ds_read");

__global__ __launch_bounds__(512) void conv3p_kernel(const Conv3pDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int b = blockIdx.x / p.tiles;
  const int ho0 = (blockIdx.x - b * p.tiles) * R;

  // this wave's first weights
  const u32x4* wsrc = p.W + (size_t)wave * WSTEPS * 128 + lane;
  u32x4 wr[WDEPTH][2];
#pragma unroll
  for (int s = 0; s < WDEPTH; ++s) {
    wr[s][0] = wsrc[s * 128];
    wr[s][1] = wsrc[s * 128 + 64];
  }

  // Beside tap t of every chunk this wave moves slots 16 (wave + 8 t) .. + 15 of a later chunk, lane (g, r) the 16-byte
  // piece g of slot r of them.  Slot s = 21 pr + 11 parity + half column is input row 2 ho0 - 1 + pr, padded column pc
  // (input column pc - 1).  The loads go through a buffer descriptor of the utterance's image: padding slots get an
  // offset far outside it and load zeros.
  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(p.in + (size_t)b * p.Hin * WIN * C), 0, p.Hin * WIN * C * 2, 0x00020000);
  const int hi0 = 2 * ho0 - 1;
  auto load = [&](int t, int cc) {
    int rr = r;
    asm volatile("" : "+v"(rr));       // worked out again at every use: nine offsets kept across the main loop would be spilled
    const int s = 16 * (wave + 8 * t) + rr;
    const int pr = s / ROW_SLOTS, rem = s - ROW_SLOTS * pr;
    const int par = rem >= 11 ? 1 : 0;
    const int pc = 2 * (rem - 11 * par) + par;
    const int hi = hi0 + pr;
    const bool valid = pr < PROWS && pc >= 1 && hi >= 0 && hi < p.Hin;
    const uint32_t off = valid ? (uint32_t)((hi * WIN + pc - 1) * (C * 2) + 16 * g) : 0xF0000000u;
#ifdef FCENC_C3_L2LOAD                                      // (timing experiment, wrong results: every load hits the L2)
    return __builtin_amdgcn_raw_buffer_load_b128(img, (int)(valid ? off & 0xFFFFu : off), 64 * cc, 0);
#else
    return __builtin_amdgcn_raw_buffer_load_b128(img, (int)off, 64 * cc, 0);
#endif
  };
  char* const wbase = smem + g * (2 * PLANE) + 256 * wave + 16 * r;    // + 2048 t + buffer: where this lane's piece goes
  {
    u32x4 first[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) first[t] = load(t, 0);
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
      if (wave + 8 * t < NGROUP) *reinterpret_cast<u32x4*>(wbase + 2048 * t) = first[t];
  }
  u32x4 ld[LDD];
#pragma unroll
  for (int t = 0; t < LDD; ++t) ld[t] = load(t, 1 + t / NTAP);

  // LDS address of tap (0, 0) of this lane's pixel of column group j in buffer 0: pixel 16 j + r is output row
  // l = (16 j + r) / 10, column wo, i.e. slot 42 l + wo = 16 j + r + 32 l.  With q = 16 j / 10 rounded down and
  // o = 16 j mod 10 (0, 6, 2, 8, 4 for j mod 5 = 0 .. 4), l is q, q + 1 or q + 2 as r passes 10 - o and 20 - o: five address
  // registers and a compile-time offset 256 j + 512 q cover all 15 groups.
  uint32_t pbase[5];
  {
    const uint32_t a = (uint32_t)(uintptr_t)smem + (uint32_t)(g * (2 * PLANE) + 16 * r);
    pbase[0] = a + (r >= 10 ? 512u : 0u);
    pbase[1] = a + (r >= 4 ? 512u : 0u) + (r >= 14 ? 512u : 0u);
    pbase[2] = a + (r >= 8 ? 512u : 0u);
    pbase[3] = a + (r >= 2 ? 512u : 0u) + (r >= 12 ? 512u : 0u);
    pbase[4] = a + (r >= 6 ? 512u : 0u);
  }
  f32x4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  u32x4 pf[RING];
  // one chunk: its 135 fragments (tap-major) through the ring, the weights of tap + WDEPTH requested as a tap ends, one
  // group of the next chunk written and the group LDD taps on requested beside every tap
  auto chunk = [&](auto bufc, const int cc) {
    constexpr int buf = decltype(bufc)::value;
    auto issue = [&](auto nc) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n < NFRAG) {
        constexpr int j = n % NJ;
        frag_read<buf * PLANE + tap_offset(n / NJ) + group_offset(j)>(pf[n % RING], pbase[j % 5]);
      }
    };
#ifndef FCENC_C3_NOREAD
    static_for<0, LEAD>([&](auto nc) { issue(nc); });
#endif
    static_for<0, NTAP>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int ws = (buf * NTAP + tap) % WDEPTH;     // ring slot of this tap's weights (chunks come in pairs)
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int n = tap * NJ + j;
#ifndef FCENC_C3_NOREAD                                     // (timing experiment, wrong results: MFMAs on stale registers)
        issue(std::integral_constant<int, n + LEAD>{});
        frag_wait<(NFRAG - 1 - n < LEAD ? NFRAG - 1 - n : LEAD)>(pf[n % RING]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma(acc[0][j], __builtin_bit_cast(bf16x8, wr[ws][0]), __builtin_bit_cast(bf16x8, pf[n % RING]));
        mma(acc[1][j], __builtin_bit_cast(bf16x8, wr[ws][1]), __builtin_bit_cast(bf16x8, pf[n % RING]));
      });
      {
        const u32x4* src_w = wsrc + (size_t)(cc * NTAP + tap + WDEPTH) * 128;   // the last WDEPTH steps are padding
        wr[ws][0] = src_w[0];
        wr[ws][1] = src_w[64];
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef FCENC_C3_NOLOAD                                     // (timing experiment, wrong results: the patch is loaded once)
      if (cc + 1 < NCHUNK && wave + 8 * tap < NGROUP)
#ifdef FCENC_C3_NOWRITE                                     // (timing experiment, wrong results: loads waited for, not written)
        asm volatile("" :: "v"(ld[tap % LDD]));
#else
        *reinterpret_cast<u32x4*>(wbase + (buf ^ 1) * PLANE + 2048 * tap) = ld[tap % LDD];
#endif
      {
        constexpr int nt = (tap + LDD) % NTAP;             // the group written LDD taps from now
        const int nc = cc + 1 + (tap + LDD) / NTAP;
        if (nc < NCHUNK && wave + 8 * nt < NGROUP) ld[tap % LDD] = load(nt, nc);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };
  for (int cc = 0; cc < NCHUNK; cc += 2) {
    chunk(std::integral_constant<int, 0>{}, cc);
    chunk(std::integral_constant<int, 1>{}, cc + 1);
  }

  // ---- epilogue: bias + SiLU, bf16 tile [240 pixels][256 channels] in LDS (16-byte piece c of pixel x at c ^ (x & 31)),
  // then whole 512-byte pixel rows to global memory: the tile is one contiguous block of the output ----
  char* tile = smem;
  int lane_e = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_e));        // worked out again here: kept across the main loop it would be spilled
  const int ge = lane_e >> 4, re = lane_e & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + 32 * wave + 16 * i + 4 * ge) * kLog2e;
    const int piece = 4 * wave + 2 * i + (ge >> 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int px = 16 * j + re;
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (bf16)silu_log2e(fmaf(acc[i][j][q], kLog2e, bias[q]));   // bias carries log2e
      *reinterpret_cast<bf16x4*>(tile + px * 512 + ((piece ^ (px & 31)) << 4) + (ge & 1) * 8) = o;
    }
  }
  __syncthreads();
  const int rows = p.Hout - ho0 < R ? p.Hout - ho0 : R;
  const int npx = rows * WOUT;
  char* obase = reinterpret_cast<char*>(p.out + ((size_t)b * p.HoutPad + ho0) * WOUT * C);
#pragma unroll
  for (int it = 0; it < NPX * 32 / 512; ++it) {
    const int c = tid + 512 * it;
    const int px = c >> 5, piece = c & 31;
    if (px < npx)
      *reinterpret_cast<u32x4*>(obase + (size_t)px * 512 + piece * 16) =
          *reinterpret_cast<const u32x4*>(tile + px * 512 + ((piece ^ (px & 31)) << 4));
  }
}

// weights [256][9 * 256] bf16 (k = tap * 256 + channel) -> [8 waves][WSTEPS][2][64] fragments in consumption order
__global__ __launch_bounds__(256) void conv3p_pack_kernel(const u32x4* __restrict__ W, u32x4* __restrict__ Wp) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= 8 * WSTEPS * 128) return;
  const int lane = v & 63, g = lane >> 4, r = lane & 15, i = (v >> 6) & 1;
  const int t = (v >> 7) % WSTEPS, w = (v >> 7) / WSTEPS;
  u32x4 o = u32x4{0u, 0u, 0u, 0u};
  if (t < NCHUNK * NTAP) {
    const int cc = t / NTAP, tap = t - NTAP * cc;
    const int row = 32 * w + 16 * i + r;
    o = W[((size_t)row * (NTAP * C) + tap * C + 32 * cc + 8 * g) / 8];
  }
  Wp[v] = o;
}

}  // namespace

bool conv3p_supported(int dtype, const ConvArgs& a) {
  return dtype == DT_BF16 && a.Cin == C && a.Cout == C && a.Win == WIN && a.Wout == WOUT;
}
size_t conv3p_packed_bytes() { return (size_t)8 * WSTEPS * 128 * 16; }

int launch_conv3p_pack(const void* W, void* Wp, int channels, hipStream_t stream) {
  if (channels != C) return fail("conv3p_pack: built for 256 channels, got %d", channels);
  if (!W || !Wp) return fail("conv3p_pack: null operand");
  if (((uintptr_t)W | (uintptr_t)Wp) & 15) return fail("conv3p_pack: operands must be 16-byte aligned");
  const int n = 8 * WSTEPS * 128;
  hipLaunchKernelGGL(conv3p_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const u32x4*)W, (u32x4*)Wp);
  return check_launch("conv3p_pack");
}

int launch_conv3p(const ConvArgs& a, const void* Wp, hipStream_t stream) {
  if (!conv3p_supported(DT_BF16, a)) return fail("conv3p: built for bf16, 256 channels and 20 input columns, got %d -> %d channels, %d columns", a.Cin, a.Cout, a.Win);
  if (a.B <= 0 || a.Hin <= 0) return fail("conv3p: empty input");
  if (a.Hout != (a.Hin + 1) / 2 || a.HoutPad < a.Hout) return fail("conv3p: %d output rows (pad %d) do not match %d input rows", a.Hout, a.HoutPad, a.Hin);
  if (!a.in || !Wp || !a.bias || !a.out) return fail("conv3p: null operand");
  if (((uintptr_t)a.in | (uintptr_t)Wp | (uintptr_t)a.out | (uintptr_t)a.bias) & 15) return fail("conv3p: operands must be 16-byte aligned");
  if ((long long)a.Hin * WIN * C * 2 >= (1ll << 31)) return fail("conv3p: %d input rows per utterance are too many", a.Hin);
  Conv3pDev d{};
  d.in = (const bf16*)a.in; d.W = (const u32x4*)Wp; d.bias = a.bias; d.out = (bf16*)a.out;
  d.Hin = a.Hin; d.Hout = a.Hout; d.HoutPad = a.HoutPad; d.tiles = (a.Hout + R - 1) / R;
  if ((long long)a.B * d.tiles >= (1ll << 31)) return fail("conv3p: batch too large");
  if (reserve_lds(reinterpret_cast<const void*>(conv3p_kernel), LDS_BYTES) != hipSuccess)
    return fail("conv3p: cannot reserve %d bytes of LDS", LDS_BYTES);
  hipLaunchKernelGGL(conv3p_kernel, dim3(a.B * d.tiles), dim3(512), LDS_BYTES, stream, d);
  return check_launch("conv3p");
}

}  // namespace fcenc
