// Cached streaming step of the encoder blocks (model.py:354-364 attention cache, :222-249 conv
// cache): the two operations that carry state from chunk to chunk.  A chunk is a handful of frames
// (Tq <= 16) per stream against up to `cap` cached frames, so both kernels are bound by reading the
// caches, not by arithmetic; they use plain fp32 FMAs.  Tokens are packed: row = stream * Tq + i.
//
// Attention cache layout: K and V as rings [B][cap][HD] in the compute dtype, frame t in slot t % cap;
// all streams have seen the same number of frames (chunks arrive in lock step).
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
  T* kring = kc + (size_t)b * cap * HD + h * 64;
  T* vring = vc + (size_t)b * cap * HD + h * 64;

  // ---- queries (+ biases) to LDS; the chunk's K / V rows into their ring slots ----
  for (int e = tid; e < Tq * 64; e += 256) {
    const int i = e >> 6, dd = e & 63;
    const float q = to_f32(qrow[(size_t)i * ld + dd]);
    qu[e] = q + bias_u[h * 64 + dd];
    qv[e] = q + bias_v[h * 64 + dd];
    const int slot = (int)((frames + i) % cap);
    kring[(size_t)slot * HD + dd] = qrow[(size_t)i * ld + HD + dd];
    vring[(size_t)slot * HD + dd] = qrow[(size_t)i * ld + 2 * HD + dd];
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
  auto key_row = [&](int j, const T* ring, int col0) -> const T* {
    const long long f = first + j;
    return f >= frames ? qrow + (size_t)(f - frames) * ld + col0 : ring + (size_t)(f % cap) * HD;
  };

  // ---- scores: thread j-strided over keys, all queries per key ----
  for (int j = tid; j < kv; j += 256) {
    const T* krow = key_row(j, kring, HD);
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
    const float v = to_f32(key_row(j, vring, 2 * HD)[lane]);
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

}  // namespace

int stream_max_chunk_frames() { return TQ_MAX; }
int stream_max_cache_frames() { return CAP_MAX; }

int launch_stream_attention(int dtype, const StreamAttnArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tq > TQ_MAX) return fail("stream_attention: chunk of %d frames (1..%d supported)", a.Tq, TQ_MAX);
  if (a.cap < a.Tq || a.cap > CAP_MAX) return fail("stream_attention: cache capacity %d out of range (%d..%d)", a.cap, a.Tq, CAP_MAX);
  if (a.frames < 0) return fail("stream_attention: negative frame count");
  if (!a.qkv || !a.pos || !a.bias_u || !a.bias_v || !a.kcache || !a.vcache || !a.out) return fail("stream_attention: null operand");
  if (dtype == DT_BF16) return launch_stream_attn_t<bf16>(a, stream);
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
