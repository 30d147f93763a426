// Fused relative-position multi-head self-attention (Transformer-XL form) for gfx950.
//
//   score[i][j] = ((q_i + u_h) . k_j + (q_i + v_h) . p_{j-i}) * scale        (model.py:379-403)
//   masks: key j >= klen[b]  -> -inf  (key padding, model.py:415-418)
//          window > 0: keys outside [i-window+1, i] -> -inf (model.py:406-412, applied by the
//          caller only when T' > local_context)
//   out = softmax_j(score) . v                                                (model.py:421-428)
// where p_{j-i} is row (T'-1)+(j-i) of pos_proj(pos_emb): the closed form of the reference's
// pad/view "_rel_shift" (model.py:312-325) for equal query and key lengths.  The score matrix
// and the (T', 2T'-1) position matrix are never materialised.
//
// Layout: activations padded to Tp (multiple of 64) frames per utterance; heads padded to 64
// columns.  One workgroup = 8 waves = 128 queries of one (utterance, head); wave w owns queries
// 16w..16w+15 and walks the key tiles (64 keys).  Everything is computed TRANSPOSED so that the
// query sits on the lane (col r) and keys / value dims on the accumulator rows:
//   S^T = K . (Q+u)^T            A = K tile rows,            B = (Q+u) fragments (registers)
//   R^T = Pband . (Q+v)^T        A = 80 rows of the position band, B = (Q+v) fragments
//   S^T[j][i] += R^T[j-i+15][i]  (per-lane skew through a wave-private LDS scratch)
//   O^T += V^T . P^T             A = V tile read transposed (ds_read_b64_tr_b16 for bf16),
//                                B = the probabilities straight from the S^T accumulators
// so the softmax row reductions are in-lane plus two cross-lane steps, and P never leaves
// registers.  The band for a (query block q0, key tile j0) is rows rb..rb+191 of the position
// table with rb = (T'-1) + j0 - q0 - 127; wave w uses band rows 112-16w .. 191-16w.
//
// Staging: K, V and the band of key tile jt+1 are fetched into registers before tile jt is
// computed and written to LDS afterwards, so the L2 latency of a tile hides behind the previous
// tile's MFMAs and softmax.  The kernel is templated on one or two LDS stages; with one stage
// (two barriers per tile) a bf16 workgroup needs exactly 80 KiB and <= 128 registers, so two
// workgroups share a CU and fill each other's barrier and softmax gaps, which measured faster
// than one double-buffered workgroup per CU.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace fcenc {

constexpr int RS = 80;   // floats per query row of the skew scratch: the skewed reads step RS - 1 = 79
                         // floats per lane (odd, conflict-free); no padding, so that stage + scratch
                         // of a bf16 workgroup is exactly half of the CU's 160 KiB

constexpr int QB = 128;        // queries per workgroup
constexpr int NWAVE = QB / 16;
constexpr int BAND = 64 + QB;  // position rows one key tile needs for QB queries

template <typename T, int NST>
struct AttnTile {
  static constexpr int ROWB = 64 * sizeof(T);           // one tile row = 64 elements
  static constexpr int K_OFF = 0;                        // [64 keys][64]
  static constexpr int V_OFF = 64 * ROWB;                // [64 keys][64]
  static constexpr int P_OFF = 128 * ROWB;               // [BAND rows][64]
  static constexpr int ROWS = 128 + BAND;
  static constexpr int STAGE = ROWS * ROWB;
  static constexpr int R_OFF = NST * STAGE;              // NWAVE x [16][RS] fp32
  static constexpr int BYTES = R_OFF + NWAVE * 16 * RS * 4;
};

// V^T fragment for value-dim block db, key chunk kc: element j of lane (g, r) is
// V[key = 32kc + 16(j>>2) + 4g + (j&3)][dv = 16db + r]
__device__ __forceinline__ bf16x8 vt_frag(const bf16* vt, int db, int kc, int g, int r) {
  using lds_bf16x4 = __attribute__((address_space(3))) bf16x4;
  bf16x8 f;
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    const int key = 32 * kc + 16 * hi + 4 * g + (r >> 2);
    const char* p = reinterpret_cast<const char*>(vt) + tile_off<128>(key, 2 * db + ((r & 3) >> 1)) + 8 * (r & 1);
    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
#pragma unroll
    for (int q = 0; q < 4; ++q) f[4 * hi + q] = v[q];
  }
  return f;
}
__device__ __forceinline__ f32x8 vt_frag(const float* vt, int db, int kc, int g, int r) {
  f32x8 f;
  const int dv = 16 * db + r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int key = 32 * kc + 16 * (j >> 2) + 4 * g + (j & 3);
    const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(vt) + tile_off<256>(key, dv >> 2) +
                                                    4 * (dv & 3));
    if (j < 4) f.lo[j] = v; else f.hi[j - 4] = v;
  }
  return f;
}

__device__ __forceinline__ bf16x8 make_frag(const float (&x)[8], bf16) {
  bf16x8 f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (bf16)x[i];
  return f;
}
__device__ __forceinline__ f32x8 make_frag(const float (&x)[8], float) {
  f32x8 f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f.lo[i] = x[i];
    f.hi[i] = x[4 + i];
  }
  return f;
}

// Output row of one query: lane (g, r) holds O^T[dv = 16 db + 4g + q][query r] in ot[db][q].  fp32: four 16-byte
// stores.  bf16: the four values of a block are 8 bytes; v_permlane16_swap between the packed registers of blocks
// 2p and 2p + 1 hands lane row g the neighbouring group as well (row 0: dv 32p + 0..7, row 1: 32p + 16..23,
// row 2: 32p + 8..15, row 3: 32p + 24..31), so a lane stores 16 contiguous bytes twice instead of 8 four times
// (the store tail of this kernel is bound by the number of store instructions, not by their bytes).
__device__ __forceinline__ void store_out_row(float* orow, const f32x4 (&ot)[4], float inv, int g) {
#pragma unroll
  for (int db = 0; db < 4; ++db) store4(orow + 16 * db + 4 * g, ot[db] * inv);
}
__device__ __forceinline__ void store_out_row(bf16* orow, const f32x4 (&ot)[4], float inv, int g) {
  u32x2 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    bf16x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (bf16)(ot[db][q] * inv);
    o[db] = __builtin_bit_cast(u32x2, v);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned ax = o[2 * p][0], ay = o[2 * p][1], bx = o[2 * p + 1][0], by = o[2 * p + 1][1];
    swap_odd_even_rows(ax, bx);
    swap_odd_even_rows(ay, by);
    *reinterpret_cast<u32x4*>(orow + 32 * p + 8 * (g >> 1) + 16 * (g & 1)) = u32x4{ax, ay, bx, by};
  }
}

template <typename T, int NST>
__global__ __launch_bounds__(64 * NWAVE, 2) void relpos_attn_kernel(const T* __restrict__ qkv, const T* __restrict__ pos,
                                                                 const float* __restrict__ bias_u,
                                                                 const float* __restrict__ bias_v,
                                                                 const int32_t* __restrict__ klen_all,
                                                                 T* __restrict__ out, int H, int Tp, int Tvalid,
                                                                 int window, float scale_log2e) {
  using TL = AttnTile<T, NST>;
  using Frag = typename Traits<T>::Frag;
  constexpr int ROWB = TL::ROWB;
  constexpr int EPC = Traits<T>::EPC;
  constexpr int CPR = ROWB / 16;                       // 16-byte chunks per tile row
  constexpr int NTHR = 64 * NWAVE;
  constexpr int NCH = TL::ROWS * CPR / NTHR;           // chunks each thread stages per key tile
  static_assert(TL::ROWS * CPR % NTHR == 0, "staging must divide evenly over the threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
  const int HD = H * 64, ld = 3 * HD;
  const int klen = klen_all[b];
  const int qw = q0 + 16 * wave;          // first query of this wave
  const int qi = qw + r;                  // this lane's query frame
  const bool active = qw < Tp;            // Tp may be an odd multiple of 64: the last block is half empty

  // ---- (Q + u) and (Q + v) fragments: lane (g, r) holds query r, dims 8g..8g+7 of each k32 step ----
  Frag qu[2], qv[2];
  {
    const T* qrow = qkv + ((size_t)b * Tp + (active ? qi : 0)) * ld + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float xq[8], xu[8], xv[8], bu[8], bv[8];
      load8<T>(qrow + 32 * ks + 8 * g, xq);
      load8<float>(bias_u + h * 64 + 32 * ks + 8 * g, bu);
      load8<float>(bias_v + h * 64 + 32 * ks + 8 * g, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {     // the softmax scale (and log2 e) rides in the query operand
        xu[j] = (xq[j] + bu[j]) * scale_log2e;
        xv[j] = (xq[j] + bv[j]) * scale_log2e;
      }
      qu[ks] = make_frag(xu, T{});
      qv[ks] = make_frag(xv, T{});
    }
  }

  f32x4 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // key tiles this query block can see
  int jt_lo = 0, jt_hi = (Tvalid - 1) >> 6;
  if (window > 0) {
    const int lo = q0 - window + 1;
    jt_lo = lo > 0 ? lo >> 6 : 0;
    const int last_q = (q0 + QB - 1) >> 6;
    if (last_q < jt_hi) jt_hi = last_q;
  }
  {
    const int last = (klen - 1) >> 6;    // tiles at or beyond klen are fully masked
    if (last < jt_hi) jt_hi = last;
  }

  float* rs = reinterpret_cast<float*>(smem + TL::R_OFF) + wave * 16 * RS;
  const int ob = QB - 16 - 16 * wave;

  // ---- staging: chunk i of this thread is 16 bytes of row (tid + NTHR*i) / CPR of the stage ----
  // Source pointers and LDS offsets are formed once; a key tile later every pointer moves 64 rows on
  // (chunk i of a thread is always a K, a V or a band chunk: NTHR / CPR rows per i divide 64).  Only
  // the half-empty last block of an odd Tp / 64 can reach in front of the position table (rows its
  // idle waves would have used); it recomputes and clamps its band rows every tile instead.
  constexpr int RPI = NTHR / CPR;            // rows covered by one chunk index i
  static_assert(64 % RPI == 0, "a chunk index must not straddle the K / V / band regions");
  u32x4 pre[NCH];
  const T* src[NCH];
  uint32_t dst[NCH];
  const bool clamp_band = q0 + QB > Tp;
  const int rb0 = (Tvalid - 1) + jt_lo * 64 - q0 - (QB - 1) + 64;   // table row of band row 0 (table offset 64)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + NTHR * i;
    const int row = c / CPR, ch = c - row * CPR;
    dst[i] = tile_off<ROWB>(row, ch);
    if (RPI * i < 64) src[i] = qkv + ((size_t)b * Tp + jt_lo * 64 + row) * ld + HD + h * 64 + ch * EPC;
    else if (RPI * i < 128) src[i] = qkv + ((size_t)b * Tp + jt_lo * 64 + row - 64) * ld + 2 * HD + h * 64 + ch * EPC;
    else {
      const int prow = rb0 + row - 128;
      src[i] = pos + (size_t)(prow > 0 ? prow : 0) * HD + h * 64 + ch * EPC;
    }
  }
  const size_t kv_step = (size_t)64 * ld, band_step = (size_t)64 * HD;
  int tiles_fetched = 0;
  auto fetch = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      pre[i] = *reinterpret_cast<const u32x4*>(src[i]);
      if (RPI * i < 128) src[i] += kv_step;
      else if (!clamp_band) src[i] += band_step;
    }
    ++tiles_fetched;
    if (clamp_band) {
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (RPI * i >= 128) {
          const int c = tid + NTHR * i;
          const int row = c / CPR, ch = c - row * CPR;
          const int prow = rb0 + 64 * tiles_fetched + row - 128;
          src[i] = pos + (size_t)(prow > 0 ? prow : 0) * HD + h * 64 + ch * EPC;
        }
    }
  };
  auto commit = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(smem + stage * TL::STAGE + dst[i]) = pre[i];
  };

  if (jt_lo <= jt_hi) {
    fetch();
    commit(0);
  }
  for (int jt = jt_lo; jt <= jt_hi; ++jt) {
    const int j0 = jt * 64;
    const int stage = NST == 2 ? (jt - jt_lo) & 1 : 0;
    __syncthreads();     // tile jt is in place; with two stages, nobody reads the other stage any more
    const bool more = jt < jt_hi;
#ifndef FCENC_ATT_NOSTAGE
    if (more) fetch();
#endif

    // a wave skips tiles its 16 queries cannot see (local attention); it still stages and synchronises
    const bool sees = active && (window <= 0 || (j0 <= qw + 15 && j0 + 63 >= qw - window + 1));
    if (sees) {
    const char* kt = smem + stage * TL::STAGE + TL::K_OFF;
    const T* vt = reinterpret_cast<const T*>(smem + stage * TL::STAGE + TL::V_OFF);
    const char* pt = smem + stage * TL::STAGE + TL::P_OFF;

    // ---- S^T = K (Q+u)^T : st[kb] lane(g,r) reg q = key 16kb+4g+q, query r ----
    f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#ifdef FCENC_ATT_NOS
        if (kb + ks > 0) { st[kb] = st[0]; continue; }
#endif
        mma(st[kb], lds_frag<T, ROWB>(kt, 16 * kb + r, ks, g), qu[ks]);
      }
    }
    // ---- R^T = Pband (Q+v)^T, 80 band rows starting at ob ----
#ifndef FCENC_ATT_NOBAND
#pragma unroll
    for (int cb = 0; cb < 5; ++cb) {
      f32x4 rt = f32x4{0.f, 0.f, 0.f, 0.f};
      const int prow = ob + 16 * cb + r;     // 0..BAND-1
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        mma(rt, lds_frag<T, ROWB>(pt, prow, ks, g), qv[ks]);
      *reinterpret_cast<f32x4*>(rs + r * RS + 16 * cb + 4 * g) = rt;   // R^T[c = 16cb+4g+q][query r]
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- skew-add, scale ----
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jl = 16 * kb + 4 * g + q;
        st[kb][q] += rs[r * RS + jl - r + 15];
      }
#endif
    // ---- mask: only tiles that stick out of the key range (or of the window of one of this wave's
    // queries) need the per-element test; one wave-uniform branch decides ----
    bool inside = j0 + 64 <= klen;
    if (window > 0) inside = inside && j0 + 63 <= qw && j0 >= qw + 15 - window + 1;
    if (!inside) {
      const int j_hi = window > 0 ? (qi < klen - 1 ? qi : klen - 1) : klen - 1;    // last visible key
      const int j_lo = window > 0 ? qi - window + 1 : 0;                            // first visible key
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + 16 * kb + 4 * g + q;
          st[kb][q] = (j > j_hi || j < j_lo) ? -INFINITY : st[kb][q];
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) tmax = fmaxf(tmax, st[kb][q]);
    __builtin_amdgcn_wave_barrier();
    tmax = rows4_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    m_run = m_new;
    float psum = 0.f;
    float pk[2][8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float pv = __builtin_amdgcn_exp2f(st[kb][q] - m_use);
        psum += pv;
        pk[kb >> 1][4 * (kb & 1) + q] = pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[db] *= alpha;
    // ---- O^T += V^T P^T ----
    Frag pf[2] = {make_frag(pk[0], T{}), make_frag(pk[1], T{})};
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
#ifdef FCENC_ATT_NOPV
        if (db + kc > 0) continue;
#endif
        mma(ot[db], vt_frag(vt, db, kc, g, r), pf[kc]);
      }
    }

    if (more) {
      if (NST == 1) __syncthreads();   // single stage: everyone is done reading before it is overwritten
#ifndef FCENC_ATT_NOSTAGE
      commit(NST == 2 ? stage ^ 1 : 0);
#endif
    }
  }
  if (!active) return;

  // ---- finish: total row sums across the four lane groups, normalise, store ----
  l_run = rows4_sum(l_run);
  // 0 visible keys -> 0 * inf = NaN, as the reference's softmax over an all -inf row.  Layout
  // padding frames (>= Tvalid) are not reference frames: they stay finite (zero) so that they
  // can never leak a NaN into real frames through 0 * NaN in a later layer's P.V product.
  const float inv = (qi >= Tvalid && l_run == 0.f) ? 0.f : 1.0f / l_run;
  T* orow = out + ((size_t)b * Tp + qi) * HD + h * 64;
  store_out_row(orow, ot, inv, g);
}


// ---------------------------------------------------------------------------------------------
// Short utterances (Tp <= 256 frames, no window): the whole key range of one (utterance, head) fits in LDS at once
// -- K and V of NKT = Tp / 64 key tiles plus ONE band of 64 NKT + 128 position rows (the per-tile bands of the
// kernel above overlap by two thirds) -- so a workgroup stages everything with a single barrier and its waves then
// walk the key tiles on their own, with no workgroup barrier, no re-staging and no band re-reads in the loop.
// Same arithmetic, operand layouts, masking and skew as relpos_attn_kernel; bf16 only.
// ---------------------------------------------------------------------------------------------
template <int NKT>
struct ResidentTile {
  static constexpr int ROWB = 128;
  static constexpr int K_OFF = 0;                          // [64 NKT keys][64]
  static constexpr int V_OFF = 64 * NKT * ROWB;
  static constexpr int P_OFF = 2 * 64 * NKT * ROWB;        // [64 NKT + QB rows][64]: band row i = table row rb0 + i
  static constexpr int NBAND = 64 * NKT + QB;
  static constexpr int ROWS = 2 * 64 * NKT + NBAND;
  static constexpr int R_OFF = ROWS * ROWB;
  static constexpr int BYTES = R_OFF + NWAVE * 16 * RS * 4;
};

template <int NKT>
__global__ __launch_bounds__(64 * NWAVE) void relpos_attn_resident_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ pos,
                                                                          const float* __restrict__ bias_u,
                                                                          const float* __restrict__ bias_v,
                                                                          const int32_t* __restrict__ klen_all,
                                                                          bf16* __restrict__ out, int H, int Tp, int Tvalid,
                                                                          float scale_log2e) {
  using T = bf16;
  using TL = ResidentTile<NKT>;
  constexpr int ROWB = TL::ROWB, CPR = 8, NTHR = 64 * NWAVE;
  constexpr int NCH = TL::ROWS * CPR / NTHR;
  static_assert(TL::ROWS * CPR % NTHR == 0, "staging must divide evenly over the threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
  const int HD = H * 64, ld = 3 * HD;
  const int klen = klen_all[b];
  const int qw = q0 + 16 * wave;
  const int qi = qw + r;
  const bool active = qw < Tp;

  // ---- stage everything: chunk i of this thread is 16 bytes of row (tid + NTHR i) / 8 ----
  const int rb0 = (Tvalid - 1) - q0 - (QB - 1) + 64;        // table row of band row 0 (the table starts 64 rows early)
  u32x4 pre[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + NTHR * i;
    const int row = c >> 3, ch = c & 7;
    const bf16* src;
    if (row < 64 * NKT) src = qkv + ((size_t)b * Tp + row) * ld + HD + h * 64 + ch * 8;
    else if (row < 128 * NKT) src = qkv + ((size_t)b * Tp + row - 64 * NKT) * ld + 2 * HD + h * 64 + ch * 8;
    else {
      const int prow = rb0 + row - 128 * NKT;                // in front of the table only for rows that idle waves own
      src = pos + (size_t)(prow > 0 ? prow : 0) * HD + h * 64 + ch * 8;
    }
    pre[i] = *reinterpret_cast<const u32x4*>(src);
  }

  // ---- (Q + u) and (Q + v) fragments while the loads are in flight ----
  bf16x8 qu[2], qv[2];
  {
    const T* qrow = qkv + ((size_t)b * Tp + (active ? qi : 0)) * ld + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float xq[8], xu[8], xv[8], bu[8], bv[8];
      load8<T>(qrow + 32 * ks + 8 * g, xq);
      load8<float>(bias_u + h * 64 + 32 * ks + 8 * g, bu);
      load8<float>(bias_v + h * 64 + 32 * ks + 8 * g, bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xu[j] = (xq[j] + bu[j]) * scale_log2e;
        xv[j] = (xq[j] + bv[j]) * scale_log2e;
      }
      qu[ks] = make_frag(xu, T{});
      qv[ks] = make_frag(xv, T{});
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + NTHR * i;
    const int row = c >> 3, ch = c & 7;
    // K, V and the band are separate swizzled tiles: the swizzle only depends on the row within its region mod 8
    *reinterpret_cast<u32x4*>(smem + tile_off<ROWB>(row, ch)) = pre[i];
  }
  __syncthreads();
  if (!active) return;

  f32x4 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  int jt_hi = (Tvalid - 1) >> 6;
  {
    const int last = (klen - 1) >> 6;        // tiles at or beyond klen are fully masked
    if (last < jt_hi) jt_hi = last;
  }
  float* rs = reinterpret_cast<float*>(smem + TL::R_OFF) + wave * 16 * RS;
  const int ob = QB - 16 - 16 * wave;

  for (int jt = 0; jt <= jt_hi; ++jt) {
    const int j0 = jt * 64;
    const char* kt = smem + TL::K_OFF + jt * 64 * ROWB;
    const T* vt = reinterpret_cast<const T*>(smem + TL::V_OFF + jt * 64 * ROWB);
    const char* pt = smem + TL::P_OFF + jt * 64 * ROWB;

    f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) mma(st[kb], lds_frag<T, ROWB>(kt, 16 * kb + r, ks, g), qu[ks]);
    }
#pragma unroll
    for (int cb = 0; cb < 5; ++cb) {
      f32x4 rt = f32x4{0.f, 0.f, 0.f, 0.f};
      const int prow = ob + 16 * cb + r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) mma(rt, lds_frag<T, ROWB>(pt, prow, ks, g), qv[ks]);
      *reinterpret_cast<f32x4*>(rs + r * RS + 16 * cb + 4 * g) = rt;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jl = 16 * kb + 4 * g + q;
        st[kb][q] += rs[r * RS + jl - r + 15];
      }
    if (j0 + 64 > klen) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + 16 * kb + 4 * g + q;
          st[kb][q] = j >= klen ? -INFINITY : st[kb][q];
        }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) tmax = fmaxf(tmax, st[kb][q]);
    __builtin_amdgcn_wave_barrier();
    tmax = rows4_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    m_run = m_new;
    float psum = 0.f;
    float pk[2][8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float pv = __builtin_amdgcn_exp2f(st[kb][q] - m_use);
        psum += pv;
        pk[kb >> 1][4 * (kb & 1) + q] = pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 4; ++db) ot[db] *= alpha;
    bf16x8 pf[2] = {make_frag(pk[0], T{}), make_frag(pk[1], T{})};
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) mma(ot[db], vt_frag(vt, db, kc, g, r), pf[kc]);
    // the next tile's R^T rows go into the same scratch: this tile's skewed reads are complete (they were waited for
    // before the softmax above used them)
  }

  l_run = rows4_sum(l_run);
  const float inv = (qi >= Tvalid && l_run == 0.f) ? 0.f : 1.0f / l_run;
  T* orow = out + ((size_t)b * Tp + qi) * HD + h * 64;
  store_out_row(orow, ot, inv, g);
}

template <int NKT>
static int launch_attn_resident(const AttnArgs& a, hipStream_t stream) {
  auto k = relpos_attn_resident_kernel<NKT>;
  const int lds = ResidentTile<NKT>::BYTES;
  if (reserve_lds(reinterpret_cast<const void*>(k), lds) != hipSuccess)
    return fail("attention: cannot reserve %d bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3((a.Tp + QB - 1) / QB, a.H, a.B), dim3(64 * NWAVE), lds, stream, (const bf16*)a.qkv,
                     (const bf16*)a.pos, a.bias_u, a.bias_v, a.klen, (bf16*)a.out, a.H, a.Tp, a.Tvalid,
                     a.scale * 1.4426950408889634f);
  return check_launch("attention");
}

template <typename T, int NST>
static int launch_attn_t(const AttnArgs& a, hipStream_t stream) {
  auto k = relpos_attn_kernel<T, NST>;
  const int lds = AttnTile<T, NST>::BYTES;
  // raising the dynamic-LDS limit is a per-device function attribute; setting it again is harmless
  if (reserve_lds(reinterpret_cast<const void*>(k), lds) != hipSuccess)
    return fail("attention: cannot reserve %d bytes of LDS", lds);
  const float log2e = 1.4426950408889634f;
  hipLaunchKernelGGL(k, dim3((a.Tp + QB - 1) / QB, a.H, a.B), dim3(64 * NWAVE), lds, stream, (const T*)a.qkv,
                     (const T*)a.pos, a.bias_u, a.bias_v, a.klen, (T*)a.out, a.H, a.Tp, a.Tvalid, a.window,
                     a.scale * log2e);
  return check_launch("attention");
}

int launch_attention(int dtype, const AttnArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.H <= 0 || a.Tp <= 0 || a.Tp % 64) return fail("attention: Tp=%d must be a positive multiple of 64", a.Tp);
  if (a.Tvalid <= 0 || a.Tvalid > a.Tp) return fail("attention: Tvalid=%d out of range", a.Tvalid);
  if (!a.qkv || !a.pos || !a.bias_u || !a.bias_v || !a.klen || !a.out) return fail("attention: null operand");
  if (a.window < 0) return fail("attention: negative window");
  // The resident variant holds one workgroup per CU (152 KiB of LDS at Tp = 256) where the tiled kernel holds two:
  // measured at small 17 L, Tp = 256, it takes 10.8 vs 12.9 us at 64 workgroups (8 utterances) but 25.7 vs 22.7 us
  // at 512 (64 utterances, two rounds), so it is used while the grid fits the chip in one round.
  // FCENC_ATTN_TILED=1 in the environment (read once per process) forces the tiled kernel, for A/B timing runs.
  static const bool force_tiled = getenv("FCENC_ATTN_TILED") != nullptr;
  const long long nwg = (long long)((a.Tp + QB - 1) / QB) * a.H * a.B;
  if (dtype == DT_BF16 && a.window == 0 && a.Tp <= 256 && nwg <= 256 && !force_tiled) {
    switch (a.Tp / 64) {
      case 1: return launch_attn_resident<1>(a, stream);
      case 2: return launch_attn_resident<2>(a, stream);
      case 3: return launch_attn_resident<3>(a, stream);
      default: return launch_attn_resident<4>(a, stream);
    }
  }
  if (dtype == DT_BF16) return launch_attn_t<bf16, 1>(a, stream);
  if (dtype == DT_F32) return launch_attn_t<float, 1>(a, stream);
  return fail("attention: dtype %d", dtype);
}

}  // namespace fcenc
