// Cached streaming step of the encoder blocks (model.py:354-364 attention cache, :222-249 conv
// cache): the two operations that carry state from chunk to chunk.  A chunk is a handful of frames
// (Tq <= 16) per stream against up to `cap` cached frames, so both are bound by reading the caches,
// not by arithmetic.  Tokens are packed: row = stream * Tq + i.
//
// Attention cache layout, per (stream, head), in the compute dtype, frame t in slot t % cap:
//   K   [B][H][cap][64]   one 128-byte (bf16) row per cached frame
//   V^T [B][H][64][cap]   one row per value dimension, cached frames along the row
// so that both can be read from global memory straight into MFMA fragments (K rows as the A operand
// of S^T = K Q^T, V^T rows as the B operand of O = P V).  All streams have seen the same number of
// frames (chunks arrive in lock step).
#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int TQ_MAX = 16;      // frames per chunk the kernels are sized for
constexpr int CAP_MAX = 512;    // cached frames (local_context) they are sized for

// ---------------------------------------------------------------------------------------------
// One workgroup = one (stream, head): append the chunk's K / V rows to the rings, then
//   score[i][j] = ((q_i + u) . k_j + shift[i][j]) * scale,   out_i = softmax_j(score[i]) . v_j
// over the kv = min(frames + Tq, cap) most recent keys.  shift is the reference's pad/view
// "_rel_shift" of bd[r][c] = (q_r + v) . p_c (model.py:312-325) followed by its fix-up for
// kv != Tq (model.py:394-400): with w = 2 Tq - 1 columns, element (i, jp) of the shifted matrix is
// element f = i w + jp + Tq of the row-major (Tq, 2 Tq) left-padded matrix, i.e. 0 when
// f % 2Tq == 0, else bd[f / 2Tq][f % 2Tq - 1]; key j takes column jp = j - (kv - w) and keys in
// front of that window take 0 (for kv < w the first kv columns are used).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stream_attn_kernel(const T* __restrict__ qkv, const T* __restrict__ pos,
                                                          const float* __restrict__ bias_u,
                                                          const float* __restrict__ bias_v, T* __restrict__ kc,
                                                          T* __restrict__ vc, T* __restrict__ out, int H, int Tq,
                                                          int cap, long long frames, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qu = reinterpret_cast<float*>(smem);          // [TQ_MAX][64]
  float* qv = qu + TQ_MAX * 64;                        // [TQ_MAX][64]
  float* sh = qv + TQ_MAX * 64;                        // [TQ_MAX][2 TQ_MAX]   shifted position scores
  float* part = sh + TQ_MAX * 2 * TQ_MAX;              // [4][TQ_MAX][64]      partial outputs
  float* sc = part + 4 * TQ_MAX * 64;                  // [TQ_MAX][cap]        scores, then probabilities

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int HD = H * 64, ld = 3 * HD, w = 2 * Tq - 1;
  const long long total = frames + Tq;
  const int kv = total < cap ? (int)total : cap;
  const long long first = total - kv;                  // absolute frame of key 0
  const T* qrow = qkv + (size_t)b * Tq * ld + h * 64;
  T* kring = kc + (size_t)(b * H + h) * cap * 64;         // [cap][64]
  T* vring = vc + (size_t)(b * H + h) * 64 * cap;         // [64][cap]

  // ---- queries (+ biases) to LDS; the chunk's K / V rows into their ring slots ----
  for (int e = tid; e < Tq * 64; e += 256) {
    const int i = e >> 6, dd = e & 63;
    const float q = to_f32(qrow[(size_t)i * ld + dd]);
    qu[e] = q + bias_u[h * 64 + dd];
    qv[e] = q + bias_v[h * 64 + dd];
    const int slot = (int)((frames + i) % cap);
    kring[(size_t)slot * 64 + dd] = qrow[(size_t)i * ld + HD + dd];
    vring[(size_t)dd * cap + slot] = qrow[(size_t)i * ld + 2 * HD + dd];
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

  // the chunk's own keys / values are read from qkv, older ones from the rings (no read of a slot
  // written above is needed, so no ordering between this workgroup's global writes and reads)
  auto key_row = [&](int j) -> const T* {
    const long long f = first + j;
    return f >= frames ? qrow + (size_t)(f - frames) * ld + HD : kring + (size_t)(f % cap) * 64;
  };
  auto value = [&](int j, int dd) -> float {
    const long long f = first + j;
    return to_f32(f >= frames ? qrow[(size_t)(f - frames) * ld + 2 * HD + dd] : vring[(size_t)dd * cap + (int)(f % cap)]);
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
// bf16 version on the matrix cores: one WAVE = one (stream, head), four per workgroup, no workgroup
// barrier.  The wave reads its 2 * cap * 128 bytes of K and V^T once, from global memory straight into
// MFMA fragments (HBM-bound: algorithmic bytes per launch = B * H * 2 * kv * 64 * 2).
//
// Keys are visited in ring-slot order, not in time order: softmax and P V do not care, and the two
// things that do -- which slots hold one of the kv most recent frames, and the position term of the
// last min(kv, 2 Tq - 1) keys -- only need a slot's age = (newest slot - slot) mod cap.
//
// Contraction order inside the MFMAs is free as long as both operands agree:
//   * head dims: k32 step ks, lane group g, element e  <->  dim 16 g + 8 ks + e, so a lane reads 32
//     contiguous bytes of a K / Q / position row;
//   * slots: the two 16-row tiles p = 0, 1 of a 32-slot step u take row m <-> slot
//     32 u + 8 (m >> 2) + 4 p + (m & 3); the accumulator then leaves lane (g, r) with the scores of
//     query r for slots 32 u + 8 g + 0..7 -- exactly the A fragment of P for the P V step, whose B
//     fragment is the 16 contiguous bytes V^T[dim][32 u + 8 g ..].
// ---------------------------------------------------------------------------------------------
template <int NT>      // 16-slot tiles: cap <= 16 NT
__global__ __launch_bounds__(256) void stream_attn_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ pos,
                                                               const float* __restrict__ bias_u,
                                                               const float* __restrict__ bias_v, bf16* __restrict__ kc,
                                                               bf16* __restrict__ vc, bf16* __restrict__ out, int BH,
                                                               int H, int Tq, int cap, long long frames,
                                                               float scale_log2e) {
  __shared__ float bdl[4][16][33];     // per wave: bd[query][c] = (q + v) . p_c
  __shared__ float linv[4][16];        // per wave: 1 / softmax denominator per query
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= BH) return;                // whole wave; nothing below synchronises across waves
  const int b = bh / H, h = bh - b * H;
  const int HD = H * 64, ld = 3 * HD, w = 2 * Tq - 1;
  const long long total = frames + Tq;
  const int kv = total < cap ? (int)total : cap;
  const int nslot = (int)((total - 1) % cap);          // slot of the newest frame
  const int jtop = (kv < w ? kv : w) - 1;              // ages 0..jtop carry a position term
  const bf16* qrow = qkv + (size_t)b * Tq * ld + h * 64;
  bf16* kring = kc + (size_t)bh * cap * 64;
  bf16* vring = vc + (size_t)bh * 64 * cap;

  // ---- the chunk's K rows / V^T columns into their slots (lane = head dim) ----
  for (int i = 0; i < Tq; ++i) {
    const int slot = (int)((frames + i) % cap);
    kring[(size_t)slot * 64 + lane] = qrow[(size_t)i * ld + HD + lane];
    vring[(size_t)lane * cap + slot] = qrow[(size_t)i * ld + 2 * HD + lane];
  }
  // the loads below are by other lanes of this wave: order them behind the stores
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // ---- (q + u), (q + v) of query r as B fragments, softmax scale and log2(e) folded in ----
  bf16x8 qu[2], qv[2];
  {
    float q[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) q[e] = 0.f;
    const bool live = r < Tq;
    if (live) {
      float t8[8];
      load8<bf16>(qrow + (size_t)r * ld + 16 * g, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = t8[e];
      load8<bf16>(qrow + (size_t)r * ld + 16 * g + 8, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) q[8 + e] = t8[e];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float bu = bias_u[h * 64 + 16 * g + 8 * ks + e], bv = bias_v[h * 64 + 16 * g + 8 * ks + e];
        qu[ks][e] = (bf16)(live ? (q[8 * ks + e] + bu) * scale_log2e : 0.f);
        qv[ks][e] = (bf16)(live ? (q[8 * ks + e] + bv) * scale_log2e : 0.f);
      }
  }

  // ---- bd^T[c][query] = p_c . (q + v): position rows as the A operand; into the wave's LDS table ----
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (16 * t >= w) break;
    const int c = 16 * t + r;
    bf16x8 p0 = zero8, p1 = zero8;
    if (c < w) {
      const bf16* prow = pos + (size_t)c * HD + h * 64 + 16 * g;
      p0 = *reinterpret_cast<const bf16x8*>(prow);
      p1 = *reinterpret_cast<const bf16x8*>(prow + 8);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    mma(acc, p0, qv[0]);
    mma(acc, p1, qv[1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) bdl[wave][r][16 * t + 4 * g + q] = acc[q];
  }
  __builtin_amdgcn_wave_barrier();

  // ---- S^T = K (q + u)^T, tile by tile, K fragments prefetched one batch of tiles ahead ----
  constexpr int KB = NT < 4 ? NT : 4;                  // tiles per batch
  constexpr int NB = NT / KB;
  static_assert(NT % KB == 0 && NT % 2 == 0, "tiles come in pairs");
  f32x4 sc[NT];
  bf16x8 kf[2][KB][2];
  auto slot_of_row = [&](int t) { return 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3); };
#pragma unroll
  for (int nb = 0; nb <= NB; ++nb) {
    if (nb < NB) {
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        const int slot = slot_of_row(nb * KB + i);
        bf16x8 k0 = zero8, k1 = zero8;
        if (slot < cap) {
          const bf16* krow = kring + (size_t)slot * 64 + 16 * g;
          k0 = *reinterpret_cast<const bf16x8*>(krow);
          k1 = *reinterpret_cast<const bf16x8*>(krow + 8);
        }
        kf[nb & 1][i][0] = k0;
        kf[nb & 1][i][1] = k1;
      }
    }
    if (nb > 0) {
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma(acc, kf[(nb - 1) & 1][i][0], qu[0]);
        mma(acc, kf[(nb - 1) & 1][i][1], qu[1]);
        sc[(nb - 1) * KB + i] = acc;
      }
    }
  }

  // ---- position term of the most recent keys, validity, softmax over the slots (per query r) ----
  // lane (g, r), tile t, register q: slot 32 (t >> 1) + 8 g + 4 (t & 1) + q
  const float inv2tq = 1.0f / (float)(2 * Tq);
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int s0 = 32 * (t >> 1) + 8 * g + 4 * (t & 1);
    int age[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = s0 + q;
      age[q] = s < cap ? nslot - s + (s > nslot ? cap : 0) : cap;      // >= kv: not one of the kv keys
    }
    const bool near = r < Tq && (age[0] <= jtop || age[1] <= jtop || age[2] <= jtop || age[3] <= jtop);
    if (__builtin_amdgcn_ballot_w64(near)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (r < Tq && age[q] <= jtop) {
          const int f = r * w + (jtop - age[q]) + Tq;
          const int rr = (int)(((float)f + 0.5f) * inv2tq);
          const int c = f - rr * 2 * Tq;
          if (c != 0) sc[t][q] += bdl[wave][rr][c - 1];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (age[q] >= kv) sc[t][q] = -INFINITY;
      m = fmaxf(m, sc[t][q]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float p = __builtin_amdgcn_exp2f(sc[t][q] - m);
      sc[t][q] = p;
      l += p;
    }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (g == 0) linv[wave][r] = 1.0f / l;
  __builtin_amdgcn_wave_barrier();

  // ---- O = P V: P from the score registers (A operand), V^T rows from global (B operand) ----
  constexpr int NU = NT / 2;           // 32-slot steps
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 vf[2][4];
#pragma unroll
  for (int u = 0; u <= NU; ++u) {
    if (u < NU) {
      const int s = 32 * u + 8 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x8 v = zero8;
        if (s < cap) v = *reinterpret_cast<const bf16x8*>(vring + (size_t)(16 * dt + r) * cap + s);
        vf[u & 1][dt] = v;
      }
    }
    if (u > 0) {
      bf16x8 pf;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pf[q] = (bf16)sc[2 * (u - 1)][q];
        pf[4 + q] = (bf16)sc[2 * (u - 1) + 1][q];
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) mma(o[dt], pf, vf[(u - 1) & 1][dt]);
    }
  }

  // lane (g, r), register q: query 4 g + q, dim 16 dt + r
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * g + q;
    if (i < Tq) {
      const float s = linv[wave][i];
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
int launch_stream_attn_t(const StreamAttnArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)(2 * TQ_MAX * 64 + TQ_MAX * 2 * TQ_MAX + 4 * TQ_MAX * 64 + TQ_MAX * a.cap) * sizeof(float);
  auto k = stream_attn_kernel<T>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail("stream_attention: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3(a.H, a.B), dim3(256), lds, stream, (const T*)a.qkv, (const T*)a.pos, a.bias_u, a.bias_v,
                     (T*)a.kcache, (T*)a.vcache, (T*)a.out, a.H, a.Tq, a.cap, a.frames, a.scale);
  return check_launch("stream_attention");
}

template <int NT>
int launch_stream_attn_mfma(const StreamAttnArgs& a, hipStream_t stream) {
  const int BH = a.B * a.H;
  hipLaunchKernelGGL(stream_attn_mfma_kernel<NT>, dim3((BH + 3) / 4), dim3(256), 0, stream, (const bf16*)a.qkv,
                     (const bf16*)a.pos, a.bias_u, a.bias_v, (bf16*)a.kcache, (bf16*)a.vcache, (bf16*)a.out, BH, a.H,
                     a.Tq, a.cap, a.frames, a.scale * 1.4426950408889634f);
  return check_launch("stream_attention");
}

}  // namespace

int stream_max_chunk_frames() { return TQ_MAX; }
int stream_max_cache_frames() { return CAP_MAX; }

int launch_stream_attention(int dtype, const StreamAttnArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tq > TQ_MAX) return fail("stream_attention: chunk of %d frames (1..%d supported)", a.Tq, TQ_MAX);
  if (a.cap < a.Tq || a.cap > CAP_MAX) return fail("stream_attention: cache capacity %d out of range (%d..%d)", a.cap, a.Tq, CAP_MAX);
  if (a.frames < 0) return fail("stream_attention: negative frame count");
  if (!a.qkv || !a.pos || !a.bias_u || !a.bias_v || !a.kcache || !a.vcache || !a.out) return fail("stream_attention: null operand");
  if (dtype == DT_BF16) {
    // matrix-core kernel: V^T rows are read 8 slots (16 bytes) at a time
    if (a.cap % 8 == 0 && (((uintptr_t)a.qkv | (uintptr_t)a.pos | (uintptr_t)a.kcache | (uintptr_t)a.vcache) & 15) == 0) {
      if (a.cap <= 64) return launch_stream_attn_mfma<4>(a, stream);
      if (a.cap <= 256) return launch_stream_attn_mfma<16>(a, stream);
      return launch_stream_attn_mfma<32>(a, stream);
    }
    return launch_stream_attn_t<bf16>(a, stream);
  }
  if (dtype == DT_F32) return launch_stream_attn_t<float>(a, stream);
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
