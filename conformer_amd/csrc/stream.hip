// Cached streaming step of the encoder blocks (model.py:354-364 attention cache, :222-249 conv
// cache): the two operations that carry state from chunk to chunk.  A chunk is a handful of frames
// (Tq <= 16) per stream against up to `cap` cached frames, so both are bound by reading the caches,
// not by arithmetic.  Tokens are packed: row = stream * Tq + i.
//
// Attention cache layout, per (stream, head), in the compute dtype, frame t in slot t % cap:
//   K [B][H][cap][64]        one 128-byte (bf16) row per cached frame
//   V [B][H][cap/8][64][8]   blocks of 8 slots; inside a block one 16-byte (bf16) group per value
//                            dimension holding that dimension for the block's 8 slots
// Both are read from global memory in exactly the shape of an MFMA fragment (a K row piece is an A
// fragment of S^T = K Q^T, a V group is a B fragment of O = P V), and a new frame touches one line of
// K and eight lines of V.  All streams have seen the same number of frames (chunks arrive in lock
// step); cap is a multiple of 8.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int TQ_MAX = 16;      // frames per chunk the kernels are sized for
constexpr int CAP_MAX = 512;    // cached frames (local_context) they are sized for

template <typename T>
__device__ __forceinline__ size_t v_index(int slot, int dim) {
  return ((size_t)(slot >> 3) * 64 + dim) * 8 + (slot & 7);
}

// ---------------------------------------------------------------------------------------------
// Any dtype, plain FMAs.  One workgroup = one (stream, head): append the chunk's K / V rows to the
// rings, then
//   score[i][j] = ((q_i + u) . k_j + shift[i][j]) * scale,   out_i = softmax_j(score[i]) . v_j
// over the kv = min(frames + Tq, cap) most recent keys.  shift is the reference's pad/view
// "_rel_shift" of bd[r][c] = (q_r + v) . p_c (model.py:312-325) followed by its fix-up for
// kv != Tq (model.py:394-400): with w = 2 Tq - 1 columns, element (i, jp) of the shifted matrix is
// element f = i w + jp + Tq of the row-major (Tq, 2 Tq) left-padded matrix, i.e. 0 when
// f % 2Tq == 0, else bd[f / 2Tq][f % 2Tq - 1]; key j takes column jp = j - (kv - w) and keys in
// front of that window take 0 (for kv < w the first kv columns are used).
// fslot = frames % cap: the slot of the chunk's first frame.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stream_attn_kernel(const T* __restrict__ qkv, const T* __restrict__ pos,
                                                          const float* __restrict__ bias_u,
                                                          const float* __restrict__ bias_v, T* __restrict__ kc,
                                                          T* __restrict__ vc, T* __restrict__ out, int H, int Tq,
                                                          int cap, int fslot_, int kv_, const int* __restrict__ dyn,
                                                          float scale) {
  // ring position: by value, or (a captured graph replays with the same arguments) from device memory
  const int fslot = dyn ? dyn[0] : fslot_, kv = dyn ? dyn[1] : kv_;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qu = reinterpret_cast<float*>(smem);          // [TQ_MAX][64]
  float* qv = qu + TQ_MAX * 64;                        // [TQ_MAX][64]
  float* sh = qv + TQ_MAX * 64;                        // [TQ_MAX][2 TQ_MAX]   shifted position scores
  float* part = sh + TQ_MAX * 2 * TQ_MAX;              // [4][TQ_MAX][64]      partial outputs
  float* sc = part + 4 * TQ_MAX * 64;                  // [TQ_MAX][cap]        scores, then probabilities

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int HD = H * 64, ld = 3 * HD, w = 2 * Tq - 1;
  const T* qrow = qkv + (size_t)b * Tq * ld + h * 64;
  T* kring = kc + (size_t)(b * H + h) * cap * 64;
  T* vring = vc + (size_t)(b * H + h) * cap * 64;
  auto wrap = [&](int s) { return s >= cap ? s - cap : s; };

  // ---- queries (+ biases) to LDS; the chunk's K / V rows into their ring slots ----
  for (int e = tid; e < Tq * 64; e += 256) {
    const int i = e >> 6, dd = e & 63;
    const float q = to_f32(qrow[(size_t)i * ld + dd]);
    qu[e] = q + bias_u[h * 64 + dd];
    qv[e] = q + bias_v[h * 64 + dd];
    const int slot = wrap(fslot + i);
    kring[(size_t)slot * 64 + dd] = qrow[(size_t)i * ld + HD + dd];
    vring[v_index<T>(slot, dd)] = qrow[(size_t)i * ld + 2 * HD + dd];
  }
  __syncthreads();

  // ---- bd[r][c] -> shifted table sh[i][jp] ----
  for (int e = tid; e < Tq * w; e += 256) {
    const int i = e / w, jp = e - i * w;
    const int f = i * w + jp + Tq;
    const int r = f / (2 * Tq), c = f - r * 2 * Tq;
    float s = 0.f;
    if (c != 0) {
      const T* prow = pos + (size_t)(c - 1) * HD + h * 64;
      const float* q = qv + r * 64;
#pragma unroll 8
      for (int dd = 0; dd < 64; ++dd) s = fmaf(q[dd], to_f32(prow[dd]), s);
    }
    sh[i * 2 * TQ_MAX + jp] = s;
  }
  __syncthreads();

  // key j (0 = oldest of the kv keys) is the chunk's frame j - (kv - Tq) when that is >= 0: those are
  // read from qkv, older ones from the rings (no read of a slot written above is needed, so no ordering
  // between this workgroup's global writes and reads); the oldest key sits Tq - kv slots from fslot
  const int first_slot = fslot + Tq - kv + (fslot + Tq - kv < 0 ? cap : 0);
  auto key_row = [&](int j) -> const T* {
    const int i = j - (kv - Tq);
    return i >= 0 ? qrow + (size_t)i * ld + HD : kring + (size_t)wrap(first_slot + j) * 64;
  };
  auto value = [&](int j, int dd) -> float {
    const int i = j - (kv - Tq);
    return to_f32(i >= 0 ? qrow[(size_t)i * ld + 2 * HD + dd] : vring[v_index<T>(wrap(first_slot + j), dd)]);
  };

  // ---- scores: thread j-strided over keys, all queries per key ----
  for (int j = tid; j < kv; j += 256) {
    const T* krow = key_row(j);
    float k[64];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      float t8[8];
      load8<T>(krow + 8 * c8, t8);
#pragma unroll
      for (int q = 0; q < 8; ++q) k[8 * c8 + q] = t8[q];
    }
    const int jp = kv >= w ? j - (kv - w) : j;
    for (int i = 0; i < Tq; ++i) {
      float s = 0.f;
      const f32x4* q = reinterpret_cast<const f32x4*>(qu + i * 64);
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        const f32x4 qq = q[c4];
        s = fmaf(qq[0], k[4 * c4], fmaf(qq[1], k[4 * c4 + 1], fmaf(qq[2], k[4 * c4 + 2], fmaf(qq[3], k[4 * c4 + 3], s))));
      }
      if (jp >= 0) s += sh[i * 2 * TQ_MAX + jp];
      sc[i * cap + j] = s * scale;
    }
  }
  __syncthreads();

  // ---- softmax per query (one wave per query) ----
  for (int i = wave; i < Tq; i += 4) {
    float m = -INFINITY;
    for (int j = lane; j < kv; j += 64) m = fmaxf(m, sc[i * cap + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float l = 0.f;
    for (int j = lane; j < kv; j += 64) {
      const float p = __expf(sc[i * cap + j] - m);
      sc[i * cap + j] = p;
      l += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    const float inv = 1.0f / l;
    for (int j = lane; j < kv; j += 64) sc[i * cap + j] *= inv;
  }
  __syncthreads();

  // ---- out_i = sum_j p_ij v_j: lane = value dim, waves split the keys ----
  float acc[TQ_MAX];
#pragma unroll
  for (int i = 0; i < TQ_MAX; ++i) acc[i] = 0.f;
  for (int j = wave; j < kv; j += 4) {
    const float v = value(j, lane);
#pragma unroll
    for (int i = 0; i < TQ_MAX; ++i)
      if (i < Tq) acc[i] = fmaf(sc[i * cap + j], v, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < TQ_MAX; ++i)
    if (i < Tq) part[(wave * TQ_MAX + i) * 64 + lane] = acc[i];
  __syncthreads();
  for (int e = tid; e < Tq * 64; e += 256) {
    const int i = e >> 6, dd = e & 63;
    const float o = part[(0 * TQ_MAX + i) * 64 + dd] + part[(1 * TQ_MAX + i) * 64 + dd] +
                    part[(2 * TQ_MAX + i) * 64 + dd] + part[(3 * TQ_MAX + i) * 64 + dd];
    out[((size_t)b * Tq + i) * HD + h * 64 + dd] = (T)o;
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 on the matrix cores: one WAVE = one (stream, head), four per workgroup, no workgroup barrier.
// The wave reads its 2 * cap * 128 bytes of K and V once (HBM-bound: algorithmic bytes per launch =
// B * H * 2 * cap * 64 * 2) through a per-wave FIFO in LDS: every 16-byte-per-lane fragment load is a
// direct-to-LDS load into one of FIFO_DEPTH 1-KiB entries (lane l's 16 bytes land at entry + 16 l and
// are read back by the same lane, so the FIFO has no layout).  The wave so keeps FIFO_DEPTH KiB of
// reads in flight through its softmax phase and across the K -> V switch without spending registers.
// Entry order = use order: K tiles 0..NT-1 (two entries each), then V steps 0..NT/2-1 (four each); a
// group is four entries; vmcnt counts the loads, so "group gi has landed" is a constant count of
// younger loads.
//
// Keys are visited in ring-slot order, not in time order: softmax and P V do not care, and the two
// things that do -- which slots hold one of the kv most recent frames, and the position term of the
// last min(kv, 2 Tq - 1) keys -- only need a slot's age = (newest slot - slot) mod cap.
//
// Contraction order inside the MFMAs is free as long as both operands agree:
//   * head dims: k32 step ks, lane group g, element e  <->  dim 16 g + 8 ks + e, so a lane reads 32
//     contiguous bytes of a K / Q / position row;
//   * slots: the two 16-row tiles p = 0, 1 of 32-slot step u take row m <-> slot
//     64 (u >> 1) + 16 (m >> 2) + 8 (u & 1) + 4 p + (m & 3); the accumulator then leaves lane (g, r)
//     with the scores of query r for the 8 slots from 64 (u >> 1) + 16 g + 8 (u & 1) on -- exactly the A
//     fragment of P for the P V step, whose B fragment is one 16-byte group of V.
//
// FULL (cap == 16 NT): every fragment address is a lane base + a block offset + a constant.  With four
// or more 64-slot blocks the blocks are visited starting behind the newest one, so that the FIFO's
// first loads (two blocks of K) touch no slot this chunk writes and can be issued before the chunk's
// rows are stored; the stores' round trip then overlaps the first loads' instead of preceding it.
// ---------------------------------------------------------------------------------------------
constexpr int FIFO_DEPTH = 16;                      // entries (KiB) per wave
constexpr int FIFO_GROUPS = FIFO_DEPTH / 4;
constexpr int FIFO_LDS = 4 * FIFO_DEPTH * 1024;     // dynamic LDS of a 4-wave workgroup

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int N>
__device__ __forceinline__ void wait_loads() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int IMM>
__device__ __forceinline__ void fifo_read(bf16x8& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

template <int NT, bool FULL>
__global__ __launch_bounds__(256) void stream_attn_fifo_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ pos,
                                                               const float* __restrict__ bias_u,
                                                               const float* __restrict__ bias_v, bf16* __restrict__ kc,
                                                               bf16* __restrict__ vc, bf16* __restrict__ out, int BH,
                                                               int H, int Tq, int cap, int fslot_, int kv_,
                                                               const int* __restrict__ dyn, float scale_log2e) {
  // ring position: by value, or (a captured graph replays with the same arguments) from device memory
  const int fslot = dyn ? __builtin_amdgcn_readfirstlane(dyn[0]) : fslot_;
  const int kv = dyn ? __builtin_amdgcn_readfirstlane(dyn[1]) : kv_;
  extern __shared__ __attribute__((aligned(1024))) char fifo_mem[];
  __shared__ float bdl[4][16][33];     // per wave: bd[query][c] = (q + v) . p_c
  static_assert(NT % 4 == 0, "tiles come in blocks of four (64 slots)");
  constexpr int NB = NT / 4;           // 64-slot blocks
  constexpr int NG = NT;               // NT/2 groups of K (two tiles each) + NT/2 groups of V (one step each)
  constexpr bool ROT = FULL && NB >= 4 && (NB & (NB - 1)) == 0;
  static_assert(!ROT || 2 * FIFO_GROUPS <= NT - 8, "the first loads must stay clear of the last two blocks");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= BH) return;                // whole wave; nothing below synchronises across waves
  const int b = bh / H, h = bh - b * H;
  const int HD = H * 64, ld = 3 * HD, w = 2 * Tq - 1;
  const int nslot = fslot + Tq - 1 - (fslot + Tq - 1 >= cap ? cap : 0);      // slot of the newest frame
  const int jtop = (kv < w ? kv : w) - 1;              // ages 0..jtop carry a position term
  const bf16* qrow = qkv + (size_t)b * Tq * ld + h * 64;
  bf16* kring = kc + (size_t)bh * cap * 64;
  bf16* vring = vc + (size_t)bh * cap * 64;
  char* fifo = fifo_mem + wave * (FIFO_DEPTH * 1024);
  const uint32_t fifo_rd = (uint32_t)(uintptr_t)fifo + lane * 16;

  // logical block L is ring block (L + bstart) mod NB; the chunk's frames lie in the last one or two
  const int bstart = ROT ? ((nslot >> 6) + 1) & (NB - 1) : 0;
  auto block_of = [&](int L) { return ROT ? (L + bstart) & (NB - 1) : L; };

  const char* kbase = reinterpret_cast<const char*>(kring + (16 * (r >> 2) + (r & 3)) * 64 + 16 * g);
  const char* vbase = reinterpret_cast<const char*>(vring + ((2 * g) * 64 + r) * 8);
  // Slots beyond cap (cap is not a multiple of 64) read the ring's last slots instead: their scores are
  // masked below and their probabilities are exactly 0, which only needs the data to be finite.
  auto issue = [&](auto gi_) {
    constexpr int gi = decltype(gi_)::value;
    constexpr int e0 = (gi % FIFO_GROUPS) * 4;
    if constexpr (gi < NT / 2) {
      const int s0 = 64 * block_of((2 * gi) >> 2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = 2 * gi + i;
        const int toff = 8 * ((t >> 1) & 1) + 4 * (t & 1);       // + the lane's 16 (r >> 2) + (r & 3)
        const char* krow = kbase + (s0 + toff) * 128;
        if constexpr (!FULL) {
          int slot = s0 + toff + 16 * (r >> 2) + (r & 3);
          slot = slot < cap ? slot : cap - 1;
          krow = reinterpret_cast<const char*>(kring + (size_t)slot * 64 + 16 * g);
        }
        glds16(krow, fifo + (e0 + 2 * i) * 1024);
        glds16(krow + 16, fifo + (e0 + 2 * i + 1) * 1024);
      }
    } else {
      constexpr int u = gi - NT / 2;
      const int s0 = 64 * block_of(u >> 1) + 8 * (u & 1);      // + the lane's 16 g
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char* vgrp = vbase + ((s0 >> 3) * 64 + 16 * dt) * 16;
        if constexpr (!FULL) {
          int s = s0 + 16 * g;
          s = s < cap ? s : cap - 8;
          vgrp = reinterpret_cast<const char*>(vring + v_index<bf16>(s, 16 * dt + r));
        }
        glds16(vgrp, fifo + (e0 + dt) * 1024);
      }
    }
  };
  if constexpr (ROT) static_for<0, FIFO_GROUPS>(issue);

  // ---- the chunk's K rows / V groups into their slots (lane = head dim), four frames' loads at a time ----
  for (int i0 = 0; i0 < Tq; i0 += 4) {
    bf16 kk[4], vv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j < Tq ? i0 + j : Tq - 1;
      kk[j] = qrow[(size_t)i * ld + HD + lane];
      vv[j] = qrow[(size_t)i * ld + 2 * HD + lane];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < Tq) {
        int slot = fslot + i0 + j;
        slot -= slot >= cap ? cap : 0;
        kring[(size_t)slot * 64 + lane] = kk[j];
        vring[v_index<bf16>(slot, lane)] = vv[j];
      }
    }
  }

  // ---- (q + u), (q + v) of query r as B fragments, softmax scale and log2(e) folded in ----
  bf16x8 qu[2], qv[2];
  {
    // every load unconditional and vectorised (rows beyond the chunk re-read its last row and are zeroed)
    const float live = r < Tq ? scale_log2e : 0.f;
    const bf16* qsrc = qrow + (size_t)(r < Tq ? r : Tq - 1) * ld + 16 * g;
    float q[2][8], bu[2][8], bv[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      load8<bf16>(qsrc + 8 * ks, q[ks]);
      load8<float>(bias_u + h * 64 + 16 * g + 8 * ks, bu[ks]);
      load8<float>(bias_v + h * 64 + 16 * g + 8 * ks, bv[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        qu[ks][e] = (bf16)((q[ks][e] + bu[ks][e]) * live);
        qv[ks][e] = (bf16)((q[ks][e] + bv[ks][e]) * live);
      }
  }

  // ---- bd^T[c][query] = p_c . (q + v): position rows as the A operand; into the wave's LDS table ----
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (16 * t >= w) break;
    const int c = 16 * t + r;
    const bf16* prow = pos + (size_t)(c < w ? c : w - 1) * HD + h * 64 + 16 * g;
    bf16x8 p0 = *reinterpret_cast<const bf16x8*>(prow);
    bf16x8 p1 = *reinterpret_cast<const bf16x8*>(prow + 8);
    if (c >= w) p0 = p1 = zero8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    mma(acc, p0, qv[0]);
    mma(acc, p1, qv[1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) bdl[wave][r][16 * t + 4 * g + q] = acc[q];
  }

  // every later load of a slot written above is by another lane of this wave: order them behind the
  // stores (this also waits for the ordinary loads above, so none of them is left to be waited for
  // once the FIFO runs)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if constexpr (!ROT) static_for<0, (FIFO_GROUPS < NG ? FIFO_GROUPS : NG)>(issue);

  // a group's four entries -> registers; the freed entries are refilled with group gi + FIFO_GROUPS
  auto take = [&](auto gi_, bf16x8(&f)[4]) {
    constexpr int gi = decltype(gi_)::value;
    constexpr int issued = gi + FIFO_GROUPS < NG ? gi + FIFO_GROUPS : NG;
    constexpr int e0 = (gi % FIFO_GROUPS) * 4;
    wait_loads<4 * (issued - gi - 1)>();
    fifo_read<(e0 + 0) * 1024>(f[0], fifo_rd);
    fifo_read<(e0 + 1) * 1024>(f[1], fifo_rd);
    fifo_read<(e0 + 2) * 1024>(f[2], fifo_rd);
    fifo_read<(e0 + 3) * 1024>(f[3], fifo_rd);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
    if constexpr (gi + FIFO_GROUPS < NG) issue(std::integral_constant<int, gi + FIFO_GROUPS>{});
  };

  // ---- S^T = K (q + u)^T ----
  f32x4 sc[NT];
  static_for<0, NT / 2>([&](auto gi_) {
    constexpr int gi = decltype(gi_)::value;
    bf16x8 f[4];
    take(gi_, f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mma(acc, f[2 * i], qu[0]);
      mma(acc, f[2 * i + 1], qu[1]);
      sc[2 * gi + i] = acc;
    }
  });

  // ---- position term of the most recent keys, validity, softmax over the slots (per query r) ----
  // lane (g, r), tile t, register q: slot 64 block_of(t >> 2) + 16 g + 8 ((t >> 1) & 1) + 4 (t & 1) + q
  const float inv2tq = 1.0f / (float)(2 * Tq);
  const int lo = nslot - jtop;         // the slots nslot - jtop .. nslot (cyclic) carry a position term
  const int abase = nslot - 16 * g;
  auto age_of = [&](int s0, int t, int q) {    // (newest slot - slot) mod cap; cap for slots beyond the ring
    const int toff = s0 + 8 * ((t >> 1) & 1) + 4 * (t & 1) + q;
    const int a = abase - toff;        // nslot - slot
    if constexpr (FULL) return a < 0 ? a + cap : a;
    return toff + 16 * g < cap ? (a < 0 ? a + cap : a) : cap;
  };
  static_for<0, NB>([&](auto blk_) {
    constexpr int L = decltype(blk_)::value;
    const int s0 = 64 * block_of(L);
    // wave-uniform: does this 64-slot block hold one of those slots?
    const bool near = lo >= 0 ? (lo <= s0 + 63 && nslot >= s0) : (nslot >= s0 || lo + cap <= s0 + 63);
    if (near) {
#pragma unroll
      for (int t = 4 * L; t < 4 * L + 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int age = age_of(s0, t, q);
          if (r < Tq && age <= jtop) {
            const int f = r * w + (jtop - age) + Tq;
            const int rr = (int)(((float)f + 0.5f) * inv2tq);
            const int c = f - rr * 2 * Tq;
            if (c != 0) {
              float bdv;     // asm: an ordinary LDS read would be made to wait for the FIFO's loads in flight
              asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                           : "=v"(bdv)
                           : "v"((uint32_t)(uintptr_t)&bdl[wave][rr][c - 1])
                           : "memory");
              sc[t][q] += bdv;
            }
          }
        }
    }
  });
  if (!(FULL && kv == cap)) {          // until the ring is full: slots that hold none of the kv keys
    static_for<0, NB>([&](auto blk_) {
      constexpr int L = decltype(blk_)::value;
      const int s0 = 64 * block_of(L);
#pragma unroll
      for (int t = 4 * L; t < 4 * L + 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (age_of(s0, t, q) >= kv) sc[t][q] = -INFINITY;
    });
  }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) m = fmaxf(fmaxf(m, sc[t][0]), fmaxf(fmaxf(sc[t][1], sc[t][2]), sc[t][3]));
  m = rows4_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float p = __builtin_amdgcn_exp2f(sc[t][q] - m);
      sc[t][q] = p;
      l += p;
    }
  l = rows4_sum(l);
  const float linv = 1.0f / l;         // of query r, in every lane group

  // ---- O = P V: P from the score registers (A operand), V groups from the FIFO (B operand) ----
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_for<0, NT / 2>([&](auto u_) {
    constexpr int u = decltype(u_)::value;
    bf16x8 f[4];
    take(std::integral_constant<int, NT / 2 + u>{}, f);
    bf16x8 pf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pf[q] = (bf16)sc[2 * u][q];
      pf[4 + q] = (bf16)sc[2 * u + 1][q];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) mma(o[dt], pf, f[dt]);
  });

  // lane (g, r), register q: query 4 g + q, dim 16 dt + r; that query's 1 / l sits in lane 4 g + q
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * g + q;
    const float s = __shfl(linv, i);
    if (i < Tq) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) out[((size_t)b * Tq + i) * HD + h * 64 + 16 * dt + r] = (bf16)(o[dt][q] * s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Causal depthwise conv (kernel 9, BatchNorm folded) + SiLU over [cache (8 frames) | chunk (Tq frames)]
// per stream, and the cache for the next chunk = the last 8 frames of that sequence
// (model.py:236-249).  A thread owns 8 channels of one stream.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stream_dwconv_kernel(const T* __restrict__ g, T* __restrict__ cache,
                                                            const float* __restrict__ w9,
                                                            const float* __restrict__ bias, T* __restrict__ out,
                                                            int B, int Tq, int d) {
  const int groups = d >> 3;
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= B * groups) return;
  const int b = item / groups, c0 = (item - b * groups) * 8;
  float w[9][8], bb[8];
#pragma unroll
  for (int k = 0; k < 9; ++k) load8<float>(w9 + (size_t)k * d + c0, w[k]);
  load8<float>(bias + c0, bb);
  T* crow = cache + (size_t)b * 8 * d + c0;
  const T* grow = g + (size_t)b * Tq * d + c0;
  // frame n of the sequence [cache | chunk]
  auto frame = [&](int n, float (&v)[8]) {
    if (n < 8) load8<T>(crow + (size_t)n * d, v);
    else load8<T>(grow + (size_t)(n - 8) * d, v);
  };
  for (int i = 0; i < Tq; ++i) {
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = bb[q];
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      float x[8];
      frame(i + m, x);
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = fmaf(w[m][q], x[q], y[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = silu_f(y[q]);
    T* o = out + ((size_t)b * Tq + i) * d + c0;
    store4(o, f32x4{y[0], y[1], y[2], y[3]});
    store4(o + 4, f32x4{y[4], y[5], y[6], y[7]});
  }
  // next cache = frames Tq .. Tq+7; row k is written after the older row Tq+k it may come from was read
  for (int k = 0; k < 8; ++k) {
    float v[8];
    frame(Tq + k, v);
    store4(crow + (size_t)k * d, f32x4{v[0], v[1], v[2], v[3]});
    store4(crow + (size_t)k * d + 4, f32x4{v[4], v[5], v[6], v[7]});
  }
}

template <typename T>
int launch_stream_attn_t(const StreamAttnArgs& a, int fslot, int kv, hipStream_t stream) {
  const size_t lds = (size_t)(2 * TQ_MAX * 64 + TQ_MAX * 2 * TQ_MAX + 4 * TQ_MAX * 64 + TQ_MAX * a.cap) * sizeof(float);
  auto k = stream_attn_kernel<T>;
  if (reserve_lds(reinterpret_cast<const void*>(k), (int)lds) != hipSuccess)
    return fail("stream_attention: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3(a.H, a.B), dim3(256), lds, stream, (const T*)a.qkv, (const T*)a.pos, a.bias_u, a.bias_v,
                     (T*)a.kcache, (T*)a.vcache, (T*)a.out, a.H, a.Tq, a.cap, fslot, kv, a.dyn, a.scale);
  return check_launch("stream_attention");
}

template <int NT>
int launch_stream_attn_fifo(const StreamAttnArgs& a, int fslot, int kv, hipStream_t stream) {
  const int BH = a.B * a.H;
  auto k = a.cap == 16 * NT ? stream_attn_fifo_kernel<NT, true> : stream_attn_fifo_kernel<NT, false>;
  if (reserve_lds(reinterpret_cast<const void*>(k), FIFO_LDS) != hipSuccess)
    return fail("stream_attention: cannot reserve %d bytes of LDS", FIFO_LDS);
  hipLaunchKernelGGL(k, dim3((BH + 3) / 4), dim3(256), FIFO_LDS, stream, (const bf16*)a.qkv, (const bf16*)a.pos,
                     a.bias_u, a.bias_v, (bf16*)a.kcache, (bf16*)a.vcache, (bf16*)a.out, BH, a.H, a.Tq, a.cap, fslot, kv,
                     a.dyn, a.scale * 1.4426950408889634f);
  return check_launch("stream_attention");
}

}  // namespace

// dyn = {slot of the next chunk's first frame, keys the next chunk's queries see, frames cached (saturating)}
__global__ void stream_advance_kernel(int* dyn, int Tq, int cap) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int fslot = dyn[0] + Tq;
    fslot -= fslot >= cap ? cap : 0;
    const int seen = dyn[2] + Tq < cap ? dyn[2] + Tq : cap;
    dyn[0] = fslot;
    dyn[1] = seen + Tq < cap ? seen + Tq : cap;
    dyn[2] = seen;
  }
}

int launch_stream_advance(int* dyn, int Tq, int cap, hipStream_t stream) {
  if (!dyn || Tq <= 0 || cap < Tq) return fail("stream_advance: bad arguments");
  hipLaunchKernelGGL(stream_advance_kernel, dim3(1), dim3(64), 0, stream, dyn, Tq, cap);
  return check_launch("stream_advance");
}

int stream_max_chunk_frames() { return TQ_MAX; }
int stream_max_cache_frames() { return CAP_MAX; }

int launch_stream_attention(int dtype, const StreamAttnArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tq > TQ_MAX) return fail("stream_attention: chunk of %d frames (1..%d supported)", a.Tq, TQ_MAX);
  if (a.cap < a.Tq || a.cap > CAP_MAX || a.cap % 8) return fail("stream_attention: cache capacity %d out of range (%d..%d, multiple of 8)", a.cap, a.Tq, CAP_MAX);
  if (a.frames < 0) return fail("stream_attention: negative frame count");
  if (!a.qkv || !a.pos || !a.bias_u || !a.bias_v || !a.kcache || !a.vcache || !a.out) return fail("stream_attention: null operand");
  if (((uintptr_t)a.qkv | (uintptr_t)a.pos | (uintptr_t)a.kcache | (uintptr_t)a.vcache | (uintptr_t)a.bias_u | (uintptr_t)a.bias_v) & 15)
    return fail("stream_attention: operands must be 16-byte aligned");
  const int fslot = (int)(a.frames % a.cap);
  const int kv = a.frames + a.Tq < a.cap ? (int)(a.frames + a.Tq) : a.cap;
  if (dtype == DT_BF16) {
    if (a.cap <= 64) return launch_stream_attn_fifo<4>(a, fslot, kv, stream);
    if (a.cap <= 256) return launch_stream_attn_fifo<16>(a, fslot, kv, stream);
    return launch_stream_attn_fifo<32>(a, fslot, kv, stream);
  }
  if (dtype == DT_F32) return launch_stream_attn_t<float>(a, fslot, kv, stream);
  return fail("stream_attention: dtype %d", dtype);
}

int launch_stream_dwconv(int dtype, const void* g, void* cache, const float* w9, const float* bias, void* out, int B,
                         int Tq, int d, hipStream_t stream) {
  if (B <= 0 || Tq <= 0 || Tq > TQ_MAX || d <= 0 || d % 8) return fail("stream_dwconv: bad shape B=%d Tq=%d d=%d", B, Tq, d);
  if (!g || !cache || !w9 || !bias || !out) return fail("stream_dwconv: null operand");
  const int items = B * (d / 8);
  const dim3 grid((items + 255) / 256), block(256);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(stream_dwconv_kernel<bf16>, grid, block, 0, stream, (const bf16*)g, (bf16*)cache, w9, bias, (bf16*)out, B, Tq, d);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL(stream_dwconv_kernel<float>, grid, block, 0, stream, (const float*)g, (float*)cache, w9, bias, (float*)out, B, Tq, d);
  else
    return fail("stream_dwconv: dtype %d", dtype);
  return check_launch("stream_dwconv");
}

}  // namespace fcenc
