// Greedy RNN-T decode on the device (train.py:180-211, inference.py:161-194): per utterance, for every
// encoder frame (padded frames included, as the reference), emit argmax symbols until blank or
// `max_symbols`, feeding the predictor only the last emitted token.
//
// The predictor is stateless and sees a single token (model.py:939-961 with a length-1 input), so its
// output -- and predictor_proj of it (model.py:1001) -- is a table indexed by that token, computed once
// by the caller; encoder_proj of every frame (model.py:1000) is one GEMM.  What is left per decision is
//   logits[v] = out_b[v] + sum_j out_w[v][j] * tanh(enc_proj[b][t][j] + pred_proj[last][j])
// clamped to +-clamp (model.py:1005-1006), and the FIRST index of the maximum (torch.argmax).  The
// decisions of one utterance are strictly sequential (a few hundred, each a V x dj matrix-vector
// product), utterances are independent: one workgroup per utterance, out_w resident in LDS when it
// fits, no host round trip per symbol.  Everything is fp32 like the reference.
#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int DEC_THREADS = 256;
constexpr int DEC_WAVES = DEC_THREADS / 64;
constexpr int DEC_MAX_DJ = 1024;
constexpr int DEC_MAX_V = 4096;
constexpr int DEC_W_LDS_FLOATS = 32768;    // out_w is kept in LDS up to 128 KiB

template <bool W_IN_LDS>
__global__ __launch_bounds__(DEC_THREADS) void greedy_decode_kernel(const float* __restrict__ enc_proj, int ldj,
                                                                    const float* __restrict__ pred_proj,
                                                                    const float* __restrict__ out_w,
                                                                    const float* __restrict__ out_b, int T, int dj,
                                                                    int V, int blank, int max_symbols, float clamp,
                                                                    int32_t* __restrict__ tokens, int cap,
                                                                    int32_t* __restrict__ counts,
                                                                    int32_t* __restrict__ last_io) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* h = reinterpret_cast<float*>(smem);             // [dj]   tanh(enc + pred)
  float* logit = h + DEC_MAX_DJ;                          // [V]
  float* wl = logit + DEC_MAX_V;                          // [V][dj] when W_IN_LDS
  __shared__ int s_tok;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* w = out_w;
  if constexpr (W_IN_LDS) {
    for (int e = tid; e < V * dj; e += DEC_THREADS) wl[e] = out_w[e];
    w = wl;
  }
  __syncthreads();

  // the loop starts from token 0 (train.py:191), or from the token a previous chunk ended on (inference.py:311-312)
  int last = last_io ? last_io[b] : 0;
  last = last < 0 ? 0 : (last >= V ? V - 1 : last);       // a table index: kept inside the table
  int n = 0;
  const float* ep = enc_proj + (size_t)b * T * ldj;
  int32_t* out = tokens + (size_t)b * cap;
  for (int t = 0; t < T; ++t) {
    for (int sym = 0; sym < max_symbols; ++sym) {
      const float* pp = pred_proj + (size_t)last * dj;
      for (int j = tid; j < dj; j += DEC_THREADS) h[j] = tanhf(ep[(size_t)t * ldj + j] + pp[j]);
      __syncthreads();
      // one wave per vocabulary entry in turn, lanes along dj
      for (int v = wave; v < V; v += DEC_WAVES) {
        const float* wr = w + (size_t)v * dj;
        float s = 0.f;
        for (int j = lane; j < dj; j += 64) s = fmaf(wr[j], h[j], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
          s += out_b[v];
          if (clamp > 0.f) s = fminf(fmaxf(s, -clamp), clamp);
          logit[v] = s;
        }
      }
      __syncthreads();
      if (wave == 0) {
        // maximum, then the smallest index that attains it
        float m = -INFINITY;
        for (int v = lane; v < V; v += 64) m = fmaxf(m, logit[v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        int idx = V;
        for (int v = lane; v < V; v += 64)
          if (logit[v] == m && v < idx) idx = v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o));
        if (lane == 0) s_tok = idx;
      }
      __syncthreads();
      const int tok = s_tok;
      if (tok == blank) break;
      if (tid == 0 && n < cap) out[n] = tok;
      ++n;
      last = tok;
    }
  }
  if (tid == 0) {
    counts[b] = n;
    if (last_io) last_io[b] = last;
  }
}

// ---- small vocabularies (V <= 32: the reference's 29 characters): output weights in registers -----------
// 256 lanes = 32 vocabulary rows x 8 slices of the joint width; a lane keeps its NJ weights in registers for
// the whole utterance.  Each wave forms tanh(enc + pred[last]) for the full width on its own (enc row in
// registers, prefetched one frame ahead; pred table in LDS) into a wave-private LDS row, so that part needs no
// workgroup barrier; the 8 slices of a row are summed with DPP, the 32 logits meet in LDS behind the one
// barrier of a decision and every wave takes their argmax itself (DPP + two readlanes -> the token is
// wave-uniform).  Per decision: NJ FMAs and 2 * NJ / 8 LDS accesses per lane, one barrier.
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// (value, index) of the larger value, the smaller index on equal values (torch.argmax takes the first maximum)
__device__ __forceinline__ void take_better(float& v, int& i, float ov, int oi) {
  const bool other = (ov > v) | ((ov == v) & (oi < i));    // no short circuit: stays branch-free
  v = other ? ov : v;
  i = other ? oi : i;
}

template <int CTRL>
__device__ __forceinline__ void argmax_step(float& v, int& i) {
  take_better(v, i, dpp_mov<CTRL>(v), dpp_mov_i<CTRL>(i));
}

// tanh to about 2 ulp without the library routine's branches: an odd polynomial below 0.25 (where 1 - e^-2|x|
// would cancel), (1 - e) / (1 + e) with e = e^-2|x| above (v_exp_f32 and v_rcp_f32 are 1 ulp each).  A wave forms
// a full-width row per decision, so the instruction count of this function is on every decision's critical path.
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
  const float e = __builtin_amdgcn_exp2f(ax * -2.885390081777927f);     // e^(-2|x|); underflows to 0 for large |x|
  const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  return copysignf(ax < 0.25f ? small : big, x);
}

template <int NJ>
__global__ __launch_bounds__(DEC_THREADS) void greedy_decode_reg_kernel(const float* __restrict__ enc_proj, int ldj,
                                                                        const float* __restrict__ pred_proj,
                                                                        const float* __restrict__ out_w,
                                                                        const float* __restrict__ out_b, int T, int dj,
                                                                        int V, int blank, int max_symbols, float clamp,
                                                                        int32_t* __restrict__ tokens, int cap,
                                                                        int32_t* __restrict__ counts,
                                                                        int32_t* __restrict__ last_io) {
  constexpr int DJP = 8 * NJ;        // padded joint width
  constexpr int NE = DJP / 64;       // elements of a full-width row per lane
  static_assert(NJ % 8 == 0, "slice width");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* pp = reinterpret_cast<float*>(smem);            // [V][DJP] predictor table, zero padded
  float* hw = pp + (size_t)V * DJP;                      // [waves][DJP] tanh row, one per wave
  float* lg = hw + DEC_WAVES * DJP;                      // [2][32] logits of this and the previous decision

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int row = wave * 8 + (lane >> 3), seg = lane & 7;

  for (int e = tid; e < V * DJP; e += DEC_THREADS) {
    const int v = e / DJP, j = e - v * DJP;
    pp[e] = j < dj ? pred_proj[(size_t)v * dj + j] : 0.f;
  }
  float w[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = seg * NJ + i;
    w[i] = (row < V && j < dj) ? out_w[(size_t)row * dj + j] : 0.f;
  }
  const float bias = row < V ? out_b[row] : 0.f;
  const float* ep = enc_proj + (size_t)b * T * ldj;
  float e_cur[NE], e_nxt[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int j = lane + 64 * i;
    e_cur[i] = j < dj ? ep[j] : 0.f;
    e_nxt[i] = 0.f;
  }
  __syncthreads();

  float* hrow = hw + wave * DJP;
  int last = last_io ? last_io[b] : 0;   // token 0 (train.py:191) or where the previous chunk ended
  last = last < 0 ? 0 : (last >= V ? V - 1 : last);
  int n = 0, parity = 0;
  int32_t* out = tokens + (size_t)b * cap;
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int j = lane + 64 * i;
        e_nxt[i] = j < dj ? ep[(size_t)(t + 1) * ldj + j] : 0.f;
      }
    }
    for (int sym = 0; sym < max_symbols; ++sym) {
      const float* pr = pp + last * DJP;
      float pv[NE];              // all table reads first: a read behind a write to the same LDS would wait for it
#pragma unroll
      for (int i = 0; i < NE; ++i) pv[i] = pr[lane + 64 * i];
#pragma unroll
      for (int i = 0; i < NE; ++i) pv[i] = tanh_fast(e_cur[i] + pv[i]);
#pragma unroll
      for (int i = 0; i < NE; ++i) hrow[lane + 64 * i] = pv[i];
      // the row is read back by other lanes of this wave only
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float s0 = 0.f, s1 = 0.f;
      const f32x4* hs = reinterpret_cast<const f32x4*>(hrow + seg * NJ);
#pragma unroll
      for (int q = 0; q < NJ / 8; ++q) {
        const f32x4 a = hs[2 * q], c = hs[2 * q + 1];
        s0 = fmaf(w[8 * q + 0], a[0], s0); s0 = fmaf(w[8 * q + 1], a[1], s0);
        s0 = fmaf(w[8 * q + 2], a[2], s0); s0 = fmaf(w[8 * q + 3], a[3], s0);
        s1 = fmaf(w[8 * q + 4], c[0], s1); s1 = fmaf(w[8 * q + 5], c[1], s1);
        s1 = fmaf(w[8 * q + 6], c[2], s1); s1 = fmaf(w[8 * q + 7], c[3], s1);
      }
      float s = s0 + s1;
      s += dpp_mov<0xB1>(s);     // the 8 slices of a row sit in 8 neighbouring lanes
      s += dpp_mov<0x4E>(s);
      s += dpp_mov<0x141>(s);
      s += bias;
      if (clamp > 0.f) s = fminf(fmaxf(s, -clamp), clamp);
      if (seg == 0) lg[parity * 32 + row] = row < V ? s : -INFINITY;
      __syncthreads();
      float bv = lg[parity * 32 + (lane & 31)];
      int bi = lane & 31;
      argmax_step<0xB1>(bv, bi);
      argmax_step<0x4E>(bv, bi);
      argmax_step<0x141>(bv, bi);
      argmax_step<0x140>(bv, bi);
      float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 0));
      int i0 = __builtin_amdgcn_readlane(bi, 0);
      const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 16));
      const int i1 = __builtin_amdgcn_readlane(bi, 16);
      take_better(v0, i0, v1, i1);
      parity ^= 1;
      const int tok = i0;
      if (tok == blank) break;
      if (tid == 0 && n < cap) out[n] = tok;
      ++n;
      last = tok;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) e_cur[i] = e_nxt[i];
  }
  if (tid == 0) {
    counts[b] = n;
    if (last_io) last_io[b] = last;
  }
}

}  // namespace

int launch_greedy_decode(const GreedyDecodeArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.T <= 0) return fail("greedy_decode: empty input B=%d T=%d", a.B, a.T);
  if (a.dj <= 0 || a.dj > DEC_MAX_DJ || a.ldj < a.dj) return fail("greedy_decode: joint width %d (ld %d) out of range (1..%d)", a.dj, a.ldj, DEC_MAX_DJ);
  if (a.V <= 1 || a.V > DEC_MAX_V) return fail("greedy_decode: vocabulary %d out of range (2..%d)", a.V, DEC_MAX_V);
  if (a.blank < 0 || a.blank >= a.V) return fail("greedy_decode: blank id %d outside the vocabulary", a.blank);
  if (a.max_symbols <= 0 || a.cap <= 0) return fail("greedy_decode: max_symbols / capacity must be positive");
  if (!a.enc_proj || !a.pred_proj || !a.out_w || !a.out_b || !a.tokens || !a.counts) return fail("greedy_decode: null operand");
  if (a.V <= 32) {
    // register-resident output weights: the smallest slice width that covers the joint width
    const int nj = a.dj <= 320 ? 40 : a.dj <= 512 ? 64 : a.dj <= 640 ? 80 : 128;
    const void* k = nj == 40 ? reinterpret_cast<const void*>(greedy_decode_reg_kernel<40>)
                  : nj == 64 ? reinterpret_cast<const void*>(greedy_decode_reg_kernel<64>)
                  : nj == 80 ? reinterpret_cast<const void*>(greedy_decode_reg_kernel<80>)
                             : reinterpret_cast<const void*>(greedy_decode_reg_kernel<128>);
    const size_t lds = ((size_t)(a.V + DEC_WAVES) * 8 * nj + 64) * sizeof(float);
    if (reserve_lds(k, (int)lds) != hipSuccess)
      return fail("greedy_decode: cannot reserve %zu bytes of LDS", lds);
    void* args[] = {(void*)&a.enc_proj, (void*)&a.ldj, (void*)&a.pred_proj, (void*)&a.out_w, (void*)&a.out_b,
                    (void*)&a.T, (void*)&a.dj, (void*)&a.V, (void*)&a.blank, (void*)&a.max_symbols, (void*)&a.clamp,
                    (void*)&a.tokens, (void*)&a.cap, (void*)&a.counts, (void*)&a.last};
    if (hipLaunchKernel(k, dim3(a.B), dim3(DEC_THREADS), args, lds, stream) != hipSuccess)
      return fail("greedy_decode: launch failed");
    return check_launch("greedy_decode");
  }
  const bool in_lds = (long long)a.V * a.dj <= DEC_W_LDS_FLOATS;
  const size_t lds = (size_t)(DEC_MAX_DJ + DEC_MAX_V + (in_lds ? a.V * a.dj : 0)) * sizeof(float);
  auto k = in_lds ? greedy_decode_kernel<true> : greedy_decode_kernel<false>;
  if (reserve_lds(reinterpret_cast<const void*>(k), (int)lds) != hipSuccess)
    return fail("greedy_decode: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3(a.B), dim3(DEC_THREADS), lds, stream, a.enc_proj, a.ldj, a.pred_proj, a.out_w, a.out_b, a.T,
                     a.dj, a.V, a.blank, a.max_symbols, a.clamp, a.tokens, a.cap, a.counts, a.last);
  return check_launch("greedy_decode");
}

}  // namespace fcenc
