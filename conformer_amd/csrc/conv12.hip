// The first two subsampling convolutions in one kernel (bf16 MFMA, gfx950; 80 mel bins, 256 channels):
//
//   standardised log-mel -> Conv2d(1 -> 256, 3x3, s2, p1) + BatchNorm + SiLU          (model.py:703-707, :741)
//                        -> Conv2d(256 -> 256, 3x3, s2, p1) + BatchNorm + SiLU        (model.py:709-713, :742)
//
// without the first convolution's activation ever leaving the chip.  As two kernels that activation
// ([B][T/2][40][256] bf16, 1.3 GB at 64 x 20 s) is written once and read 2.25 times (the nine taps of the
// second convolution share input pixels) and the second kernel is paced by what a CU can load, not by MFMA.
//
// A workgroup (8 waves) owns 12 output rows x 20 output columns (240 pixels = 15 MFMA column groups) x all 256
// output channels of one utterance.  Its conv2 input is a patch of 25 x 41 conv1 pixels; the 51 x 81 standardised
// input samples under that patch are staged once in LDS (as a bf16 head and a bf16 remainder, so that conv1 can
// run on the matrix cores at fp32 accuracy, see conv1_mfma_kernel in elementwise.hip).  The input channels are
// walked in 8 chunks of 32: while the MFMAs consume chunk c of the patch (9 taps x k32), the same waves produce
// chunk c + 1 (conv1 by MFMA, SiLU, bf16) into the other of two patch buffers; one barrier per chunk.
//
//   patch buffers: 4 planes (one per 16-byte piece = 8 channels) x 2 buffers x 1056 slots x 16 B.  A conv1 row is 42
//                  slots: its 21 even padded columns, then its odd ones, so that the stride-2 pixels of a tap are
//                  consecutive slots, tap and buffer are a compile-time byte offset of the read instruction, and the
//                  16 lanes x 16 B of a fragment read are 256 consecutive bytes of one plane: no bank conflicts, no
//                  swizzle.  Fragment reads are issued by hand four fragments ahead of the MFMAs that use them.
//   weights      : wave w multiplies output channels 32w .. 32w+31 for all 240 pixels (30 accumulators), so its
//                  weight fragments are its own: they stream from L2 straight into registers, packed in consumption
//                  order (conv12_pack_kernel), three taps ahead.  Per CU that is 16 KiB per tap, once.
//   staggering   : waves 4-7 produce before each tap's MFMAs, waves 0-3 after them, so that the two waves of a SIMD
//                  alternate between the matrix pipe and the VALU instead of meeting at both.
#include "common.h"
#include "frag_ring.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int C = 256;                    // channels of both convolutions
constexpr int NF = 80, F1 = 40, F2 = 20;  // mel bins, conv1 columns, conv2 columns
constexpr int R = 12;                     // conv2 output rows per workgroup
constexpr int NPX = R * F2;               // 240 output pixels
constexpr int NJ = NPX / 16;              // 15 MFMA column groups
constexpr int PROWS = 2 * R + 1;          // 25 conv1 rows under them
constexpr int ROW_SLOTS = 42;             // slots per conv1 row: padded columns 0, 2, .., 40 then 1, 3, .., 39 (+ 1 unused)
constexpr int NSLOT = 1056;               // 25 * 42 = 1050, rounded up to whole groups of 16
constexpr int NGROUP = NSLOT / 16;        // 66 production groups per chunk
constexpr int PLANE = NSLOT * 16;         // 16 896 B (a multiple of 256: every plane starts on bank 0)
constexpr int PBUF = 4 * PLANE;           // 67 584 B: one 32-channel chunk of the patch (its planes are 2 * PLANE apart)
constexpr int MEL_ROWS = 4 * R + 3;       // 51 input rows
constexpr int MEL_PITCH = 82;             // bf16 elements per staged row: columns w = -1 .. 79 and one unused
constexpr int MEL_ELEMS = MEL_ROWS * MEL_PITCH;
constexpr int MEL_PLANE = 8368;           // MEL_ELEMS * 2 = 8364 rounded up to 16
constexpr int MEL_OFF = 2 * PBUF;
constexpr int STAT_OFF = MEL_OFF + 2 * MEL_PLANE;
constexpr int TAB_OFF = STAT_OFF + 16;    // per patch slot: staged element of its tap (0, 0), bit 15 = the slot is a real conv1 pixel
constexpr int TAB_SLOTS = 80 * 16;         // groups 66 .. 79 do not exist; their entries are zero
constexpr int LDS_BYTES = TAB_OFF + TAB_SLOTS * 2;
constexpr int NCHUNK = C / 32;
constexpr int NTAP = 9;
#ifndef FCENC_C12_PRIO_LEVEL
#define FCENC_C12_PRIO_LEVEL 3            // wave priority while producing (0: unchanged, a timing experiment)
#endif
#ifndef FCENC_C12_RING
#define FCENC_C12_RING 5
#endif
constexpr int RING = FCENC_C12_RING, LEAD = RING - 1;  // pixel fragments in registers / read ahead of their MFMAs
constexpr int NFRAG = NTAP * NJ;          // 135 fragment reads per chunk
constexpr int WDEPTH = 2;                 // taps of weights per wave in registers: the one in use and the next
constexpr int WSTEPS = NCHUNK * NTAP + WDEPTH;   // packed steps per wave, the last WDEPTH are zero padding
constexpr int NPRE = (MEL_ELEMS + 511) / 512;    // input samples staged per thread
constexpr int OUT_TILE_BYTES = NPX * C * 2;
static_assert(OUT_TILE_BYTES <= 2 * PBUF, "the output tile reuses the patch buffers");
static_assert((2 * NTAP) % WDEPTH == 0, "ring slots must be compile-time constants over a pair of chunks");

// K slots 0..7 of a conv1 fragment hold taps (kh, kw) = (0,0) (0,1) (1,0) (1,1) (2,0) (2,1) (0,2) (1,2): three aligned
// 4-byte reads and one pair of 2-byte reads per pixel; tap (2,2) rides in the fourth lane group beside the bias
__device__ __constant__ int kTapOfSlot[8] = {0, 1, 3, 4, 6, 7, 2, 5};

__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
  hi = (bf16)v;
  lo = (bf16)(v - (float)hi);
}

struct Conv12Dev {
  const float* mel;        // [B][Tin][80] fp32
  const double* partial;   // [B][nsplit][2] sums of log(mel + 1e-6) and of its square
  const u32x4* A1;         // conv1 fragments [8 chunks][2][64 lanes]
  const u32x4* W2;         // conv2 fragments [8 waves][WSTEPS][2][64 lanes]
  const float* bias2;      // [256]
  bf16* out;               // [B][Hout][20][256]
  int nsplit, Tin, T1, Hout, tiles;
};

constexpr int tap_offset(int tap) {        // byte offset of tap (kh, kw) from tap (0, 0) inside a plane
  const int kh = tap / 3, kw = tap % 3;
  return (kh * ROW_SLOTS + (kw & 1) * 21 + (kw >> 1)) * 16;
}
constexpr int group_offset(int j) { return 256 * j + 1024 * (16 * j / 20); }   // see pbase in the kernel
static_assert(PLANE + tap_offset(8) + group_offset(14) < 65536, "buffer + tap + group must fit the 16-bit offset of a ds_read");
static_assert(NFRAG % RING == 0, "ring slots must be compile-time constants");

__global__ __launch_bounds__(512) void conv12_kernel(const Conv12Dev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int b = blockIdx.x / p.tiles;
  const int ho0 = (blockIdx.x - b * p.tiles) * R;
  bf16* melh = reinterpret_cast<bf16*>(smem + MEL_OFF);
  bf16* mell = reinterpret_cast<bf16*>(smem + MEL_OFF + MEL_PLANE);
  float* stat = reinterpret_cast<float*>(smem + STAT_OFF);

  // ---- the input samples under the patch: rows t = 4 ho0 - 3 + tr, columns w = tc - 1 ----
  float raw[NPRE];
  uint32_t inside = 0;
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = tid + 512 * i;
    const int tr = e / MEL_PITCH, tc = e - tr * MEL_PITCH;
    const int t = 4 * ho0 - 3 + tr, w = tc - 1;
    raw[i] = 1.f;
    if (e < MEL_ELEMS && t >= 0 && t < p.Tin && w >= 0 && w < NF) {
      raw[i] = p.mel[((size_t)b * p.Tin + t) * NF + w];
      inside |= 1u << i;
    }
  }
  // this wave's first weights (they do not depend on anything above)
  const u32x4* wsrc = p.W2 + (size_t)wave * WSTEPS * 128 + lane;
  u32x4 wr[WDEPTH][2];
#pragma unroll
  for (int s = 0; s < WDEPTH; ++s) {
    wr[s][0] = wsrc[s * 128];
    wr[s][1] = wsrc[s * 128 + 64];
  }
  if (tid < 64) {
    double s = 0.0, q = 0.0;
    if (tid < p.nsplit) {
      s = p.partial[((size_t)b * p.nsplit + tid) * 2];
      q = p.partial[((size_t)b * p.nsplit + tid) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    if (tid == 0) {
      const double n = (double)p.Tin * (double)NF;
      const double mean = s / n;
      double var = (q - s * mean) / (n - 1.0);
      if (var < 0.0) var = 0.0;
      stat[0] = (float)mean;
      stat[1] = 1.0f / ((float)sqrt(var) + 1e-6f);
    }
  }
  // where each patch slot finds its input samples, and whether it is a pixel at all: the same for all 8 chunks, so it
  // is worked out once here (slot s = 42 pr + 21 parity + half column; padded conv1 column pc = f1 + 1)
  uint16_t* tab = reinterpret_cast<uint16_t*>(smem + TAB_OFF);
  for (int s = tid; s < TAB_SLOTS; s += 512) {
    const int pr = s / ROW_SLOTS, rem = s - ROW_SLOTS * pr;
    const int par = rem >= 21 ? 1 : 0;
    const int pc = 2 * (rem - 21 * par) + par;
    const int t1 = 2 * ho0 - 1 + pr;
    const bool valid = pc >= 1 && pc <= F1 && pr < PROWS && t1 >= 0 && t1 < p.T1;
    tab[s] = valid ? (uint16_t)(0x8000 | (2 * pr * MEL_PITCH + 2 * (pc - 1))) : (uint16_t)0;
  }
  __syncthreads();
  {
    const float mean = stat[0], rstd = stat[1];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = tid + 512 * i;
      if (e < MEL_ELEMS) {
        bf16 hi, lo;
        split_bf16(((inside >> i) & 1u) ? (logf(raw[i] + 1e-6f) - mean) * rstd : 0.f, hi, lo);
        melh[e] = hi;
        mell[e] = lo;
      }
    }
  }
  __syncthreads();

  // ---- conv1 for one group of 16 patch slots: MFMA (weights A, pixels B), SiLU, bf16, into the patch planes.  In three
  // steps so that two groups can be taken through them side by side (their dependent chains then overlap) ----
  auto gather = [&](uint32_t ent) {                 // this lane's eight input samples as a B fragment
    const int e0 = ent & 0x7FFF;                    // staged element of tap (0, 0): even, so 4-byte aligned
    const bf16* base = (g == 1 ? mell : melh) + e0;
    uint32_t d0 = *reinterpret_cast<const uint32_t*>(base);
    uint32_t d1 = *reinterpret_cast<const uint32_t*>(base + MEL_PITCH);
    uint32_t d2 = *reinterpret_cast<const uint32_t*>(base + 2 * MEL_PITCH);
    uint32_t d3 = (uint32_t)*reinterpret_cast<const uint16_t*>(base + 2) |
                  ((uint32_t)*reinterpret_cast<const uint16_t*>(base + MEL_PITCH + 2) << 16);
    const uint32_t h8 = *reinterpret_cast<const uint16_t*>(melh + e0 + 2 * MEL_PITCH + 2);
    const uint32_t l8 = *reinterpret_cast<const uint16_t*>(mell + e0 + 2 * MEL_PITCH + 2);
    if (g == 3) {                                   // [xh8, xl8, xh8, 1, 1, 0, 0, 0] against [wh8, wh8, wl8, bh, bl, ..]
      d0 = h8 | (l8 << 16);
      d1 = h8 | 0x3F800000u;
      d2 = 0x00003F80u;
      d3 = 0u;
    }
    return u32x4{d0, d1, d2, d3};
  };
  auto finish = [&](int gi, uint32_t ent, const f32x4& a0, const f32x4& a1, char* dst) {
    int rr = r;
    asm volatile("" : "+v"(rr));                    // formed afresh at every use: the compiler would otherwise keep one
                                                    // write address per tap alive across the chunk loop, and spill them
    const int s = 16 * gi + rr;                     // this lane's slot; ent = tab[s]
    const bool valid = ent >> 15;                   // not padding (conv2's zero padding is of conv1's OUTPUT)
    bf16x8 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#ifdef FCENC_C12_NOSILU
      y[q] = (bf16)a0[q];                          // timing experiment (wrong results): no activation
      y[4 + q] = (bf16)a1[q];
#else
      y[q] = (bf16)silu_log2e(a0[q]);
      y[4 + q] = (bf16)silu_log2e(a1[q]);
#endif
    }
    u32x4 yo = __builtin_bit_cast(u32x4, y);
    if (!valid) yo = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(dst + g * (2 * PLANE) + s * 16) = yo;
  };
  auto produce = [&](int gi, uint32_t ent, const bf16x8& a0f, const bf16x8& a1f, char* dst) {
    const bf16x8 xf = __builtin_bit_cast(bf16x8, gather(ent));
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    mma(a0, a0f, xf);
    mma(a1, a1f, xf);
    finish(gi, ent, a0, a1, dst);
  };
  // two groups (gi and gi + 8) side by side; the scheduling barriers keep the compiler from spreading the steps further
  // (it would hold both groups' temporaries at once, and there are no registers for that)
  auto produce2 = [&](int gi, uint32_t e0, uint32_t e1, const bf16x8& a0f, const bf16x8& a1f, char* dst) {
    const bf16x8 x0 = __builtin_bit_cast(bf16x8, gather(e0));
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, gather(e1));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a00 = f32x4{0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00;
    mma(a00, a0f, x0);
    mma(a01, a1f, x0);
    mma(a10, a0f, x1);
    mma(a11, a1f, x1);
    __builtin_amdgcn_sched_barrier(0);
    finish(gi, e0, a00, a01, dst);
    finish(gi + 8, e1, a10, a11, dst);
  };

  // chunk 0 of the patch
  {
    const bf16x8 a0f = __builtin_bit_cast(bf16x8, p.A1[lane]);
    const bf16x8 a1f = __builtin_bit_cast(bf16x8, p.A1[64 + lane]);
    // nothing else runs yet: four groups at a time, so that their dependent chains (LDS reads -> MFMA -> exp2 -> rcp ->
    // LDS write) overlap instead of following one another
#ifdef FCENC_C12_SERIALPRO                                  // (timing experiment: one group after the other)
    for (int gi = wave; gi < 64; gi += 8) produce(gi, tab[16 * gi + r], a0f, a1f, smem);
#else
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int g0 = wave + 32 * k;
      uint32_t e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = tab[16 * (g0 + 8 * u) + r];
#pragma unroll
      for (int u = 0; u < 4; ++u) produce(g0 + 8 * u, e[u], a0f, a1f, smem);
    }
#endif
    if (wave + 64 < NGROUP) produce(wave + 64, tab[16 * (wave + 64) + r], a0f, a1f, smem);
  }

  // LDS address of tap (0, 0) of this lane's pixel of column group j in buffer 0: pixel 16 j + r is output row
  // l = (16 j + r) / 20, column wo, i.e. slot 84 l + wo = 16 j + r + 64 l.  With q = 16 j / 20 rounded down, l is q or
  // q + 1 depending on r >= 20 (q + 1) - 16 j, a threshold that is 20, 4, 8, 12, 16 for j mod 5 = 0 .. 4: four address
  // registers (threshold never met, met from lane 4, 8, 12 on) and a compile-time offset 256 j + 1024 q cover all 15 groups.
  uint32_t pbase[4];
  {
    const uint32_t a = (uint32_t)(uintptr_t)smem + (uint32_t)(g * (2 * PLANE) + 16 * r);
    pbase[0] = a;
    pbase[1] = a + (r >= 4 ? 1024u : 0u);
    pbase[2] = a + (r >= 8 ? 1024u : 0u);
    pbase[3] = a + (r >= 12 ? 1024u : 0u);
  }
  f32x4 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#if defined(FCENC_C12_ALLEARLY)
  const bool late = false;                 // timing experiment: no staggering
#elif defined(FCENC_C12_ODDLATE)
  const bool late = wave & 1;              // timing experiment: staggering by wave parity (waves on different SIMDs)
#else
  const bool late = wave >= 4;             // produce before the tap's MFMAs instead of after them
#endif
  u32x4 pf[RING];
  // table entry of the group this wave produces next, read a tap ahead of its use (groups wave, wave + 8, .. every chunk)
  uint32_t ent = tab[16 * wave + r];
#ifdef FCENC_C12_PAIR
  uint32_t ent1 = tab[16 * (wave + 8) + r];
#endif
  // one chunk: its 135 fragments (tap-major) through the ring, the weights of tap + WDEPTH requested as a tap ends, one
  // group of the next chunk's patch produced beside every tap
  auto chunk = [&](auto bufc, const int cc) {
    constexpr int buf = decltype(bufc)::value;
    char* nxt = smem + (buf ^ 1) * PLANE;
    const bool more = cc + 1 < NCHUNK;
    bf16x8 a0f = bf16x8{}, a1f = bf16x8{};
    if (more) {
      a0f = __builtin_bit_cast(bf16x8, p.A1[(cc + 1) * 128 + lane]);
      a1f = __builtin_bit_cast(bf16x8, p.A1[(cc + 1) * 128 + 64 + lane]);
    }
    auto issue = [&](auto nc) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n < NFRAG) {
        constexpr int j = n % NJ;
        frag_read<buf * PLANE + tap_offset(n / NJ) + group_offset(j)>(pf[n % RING], pbase[j % 5 == 4 ? 0 : j % 5]);
      }
    };
#ifndef FCENC_C12_NOMAIN
    static_for<0, LEAD>([&](auto nc) { issue(nc); });
#endif
    static_for<0, NTAP>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int ws = (buf * NTAP + tap) % WDEPTH;     // ring slot of this tap's weights (chunks come in pairs)
      const int gi = wave + 8 * tap;
#ifdef FCENC_C12_NOPROD
      const bool mine = false;                     // timing experiment (wrong results): the patch is produced once
#else
      const bool mine = more && gi < NGROUP;
#endif
#ifdef FCENC_C12_PAIR
      // two groups at every other tap (those of this tap and the next), so that their dependent chains overlap
      auto producer = [&]() {
        if constexpr (tap % 2 == 0) {
          const bool mine1 = mine && tap + 1 < NTAP && gi + 8 < NGROUP;
          if (mine1) {
            produce2(gi, ent, ent1, a0f, a1f, nxt);
          } else if (mine) {
            produce(gi, ent, a0f, a1f, nxt);
          }
        }
        if constexpr (tap % 2 == 1 || tap == NTAP - 1) {     // entries of the next producing tap (tap + 1, or 0 of the next chunk)
          constexpr int nt = (tap + 1) % NTAP;
          ent = tab[16 * (wave + 8 * nt) + r];
          ent1 = tab[16 * (wave + 8 * (nt + 1)) + r];          // (groups past the last one read zeros behind the table)
        }
      };
#else
      auto producer = [&]() {
        if (mine) produce(gi, ent, a0f, a1f, nxt);
        ent = tab[16 * (wave + 8 * ((tap + 1) % NTAP)) + r];       // (groups past the last one read zeros behind the table)
      };
#endif
      // The producer runs at raised wave priority: its VALU / LDS instructions then go ahead of the other wave's MFMAs
      // on the SIMD's issue port, which the matrix pipe (busy for 16 clocks per MFMA) does not feel (-3.5 % per launch).
      auto prioritised = [&]() {
#if FCENC_C12_PRIO_LEVEL > 0
        __builtin_amdgcn_s_setprio(FCENC_C12_PRIO_LEVEL);
        producer();
        __builtin_amdgcn_s_setprio(0);
#else
        producer();
#endif
      };
      __builtin_amdgcn_sched_barrier(0);
      if (late) prioritised();
      __builtin_amdgcn_sched_barrier(0);
#ifdef FCENC_C12_NOMAIN
      asm volatile("" :: "v"(wr[ws][0]), "v"(wr[ws][1]));   // timing experiment: no conv2 MFMAs
#else
      static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int n = tap * NJ + j;
        issue(std::integral_constant<int, n + LEAD>{});
        frag_wait<(NFRAG - 1 - n < LEAD ? NFRAG - 1 - n : LEAD)>(pf[n % RING]);
        __builtin_amdgcn_sched_barrier(0);
        mma(acc[0][j], __builtin_bit_cast(bf16x8, wr[ws][0]), __builtin_bit_cast(bf16x8, pf[n % RING]));
        mma(acc[1][j], __builtin_bit_cast(bf16x8, wr[ws][1]), __builtin_bit_cast(bf16x8, pf[n % RING]));
      });
#endif
      {
        const u32x4* src = wsrc + (size_t)(cc * NTAP + tap + WDEPTH) * 128;   // the last WDEPTH steps are padding
        wr[ws][0] = src[0];
        wr[ws][1] = src[64];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!late) prioritised();
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
  // thread coordinates worked out again, from the lane counter rather than threadIdx: anything kept for this across the
  // main loop would be spilled
  const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid_e = 64 * wave + lane_e;
  const int ge = lane_e >> 4, re = lane_e & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias2 + 32 * wave + 16 * i + 4 * ge) * kLog2e;
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
  const int npx = rows * F2;
  char* obase = reinterpret_cast<char*>(p.out + ((size_t)b * p.Hout + ho0) * F2 * C);
#pragma unroll
  for (int it = 0; it < NPX * 32 / 512; ++it) {
    const int c = tid_e + 512 * it;
    const int px = c >> 5, piece = c & 31;
    if (px < npx)
      *reinterpret_cast<u32x4*>(obase + (size_t)px * 512 + piece * 16) =
          *reinterpret_cast<const u32x4*>(tile + px * 512 + ((piece ^ (px & 31)) << 4));
  }
}

// conv1 weights [9][256] fp32 + bias [256] (BatchNorm folded) -> A1 [8][2][64] fragments; conv2 weights [256][9 * 256]
// bf16 (k = tap * 256 + channel) -> W2 [8 waves][WSTEPS][2][64] fragments in the order the kernel consumes them
__global__ __launch_bounds__(256) void conv12_pack_kernel(const float* __restrict__ w9, const float* __restrict__ b1,
                                                          const u32x4* __restrict__ W2, u32x4* __restrict__ A1p,
                                                          u32x4* __restrict__ W2p) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int lane = v & 63, g = lane >> 4, r = lane & 15, i = (v >> 6) & 1;
  if (v < NCHUNK * 128) {
    const int cc = v >> 7;
    const int ch = 32 * cc + 8 * (r >> 2) + 4 * i + (r & 3);
    bf16x8 a;
    if (g < 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        bf16 hi, lo;
        split_bf16(w9[(size_t)kTapOfSlot[k] * C + ch] * kLog2e, hi, lo);
        a[k] = g == 2 ? lo : hi;
      }
    } else {
      bf16 wh, wl, bh, bl;
      split_bf16(w9[(size_t)8 * C + ch] * kLog2e, wh, wl);
      split_bf16(b1[ch] * kLog2e, bh, bl);
      a[0] = wh; a[1] = wh; a[2] = wl; a[3] = bh; a[4] = bl;
      a[5] = a[6] = a[7] = (bf16)0.f;
    }
    A1p[v] = __builtin_bit_cast(u32x4, a);
  }
  if (v < 8 * WSTEPS * 128) {
    const int t = (v >> 7) % WSTEPS, w = (v >> 7) / WSTEPS;
    u32x4 o = u32x4{0u, 0u, 0u, 0u};
    if (t < NCHUNK * NTAP) {
      const int cc = t / NTAP, tap = t - NTAP * cc;
      const int row = 32 * w + 16 * i + r;
      o = W2[((size_t)row * (NTAP * C) + tap * C + 32 * cc + 8 * g) / 8];
    }
    W2p[v] = o;
  }
}

}  // namespace

bool conv12_supported(int dtype, int n_mels, int channels) { return dtype == DT_BF16 && n_mels == NF && channels == C; }
size_t conv12_a1_bytes() { return (size_t)NCHUNK * 128 * 16; }
size_t conv12_w2_bytes() { return (size_t)8 * WSTEPS * 128 * 16; }

int launch_conv12_pack(const float* w9, const float* b1, const void* W2, void* A1p, void* W2p, int n_mels, int channels,
                       hipStream_t stream) {
  if (!conv12_supported(DT_BF16, n_mels, channels)) return fail("conv12_pack: built for 80 mel bins and 256 channels, got %d / %d", n_mels, channels);
  if (!w9 || !b1 || !W2 || !A1p || !W2p) return fail("conv12_pack: null operand");
  const int n = 8 * WSTEPS * 128;
  hipLaunchKernelGGL(conv12_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, w9, b1, (const u32x4*)W2, (u32x4*)A1p,
                     (u32x4*)W2p);
  return check_launch("conv12_pack");
}

int launch_conv12(const float* mel, const double* partial, int nsplit, const void* A1p, const void* W2p, const float* bias2,
                  void* out, int B, int T, int n_mels, int channels, hipStream_t stream) {
  if (!conv12_supported(DT_BF16, n_mels, channels)) return fail("conv12: built for 80 mel bins and 256 channels, got %d / %d", n_mels, channels);
  if (B <= 0 || T <= 0) return fail("conv12: empty input");
  if (!mel || !partial || !A1p || !W2p || !bias2 || !out) return fail("conv12: null operand");
  if (nsplit <= 0 || nsplit > 64) return fail("conv12: nsplit=%d", nsplit);
  Conv12Dev d{};
  d.mel = mel; d.partial = partial; d.A1 = (const u32x4*)A1p; d.W2 = (const u32x4*)W2p; d.bias2 = bias2; d.out = (bf16*)out;
  d.nsplit = nsplit; d.Tin = T; d.T1 = (T + 1) / 2; d.Hout = (d.T1 + 1) / 2; d.tiles = (d.Hout + R - 1) / R;
  if ((long long)B * d.tiles >= (1ll << 31)) return fail("conv12: batch too large");
  if (reserve_lds(reinterpret_cast<const void*>(conv12_kernel), LDS_BYTES) != hipSuccess)
    return fail("conv12: cannot reserve %d bytes of LDS", LDS_BYTES);
  hipLaunchKernelGGL(conv12_kernel, dim3(B * d.tiles), dim3(512), LDS_BYTES, stream, d);
  return check_launch("conv12");
}

}  // namespace fcenc
