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
                                                                    int32_t* __restrict__ counts) {
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

  int last = blank;          // the predictor starts from the blank token (train.py:192-193)
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
  if (tid == 0) counts[b] = n;
}

}  // namespace

int launch_greedy_decode(const GreedyDecodeArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.T <= 0) return fail("greedy_decode: empty input B=%d T=%d", a.B, a.T);
  if (a.dj <= 0 || a.dj > DEC_MAX_DJ || a.ldj < a.dj) return fail("greedy_decode: joint width %d (ld %d) out of range (1..%d)", a.dj, a.ldj, DEC_MAX_DJ);
  if (a.V <= 1 || a.V > DEC_MAX_V) return fail("greedy_decode: vocabulary %d out of range (2..%d)", a.V, DEC_MAX_V);
  if (a.blank < 0 || a.blank >= a.V) return fail("greedy_decode: blank id %d outside the vocabulary", a.blank);
  if (a.max_symbols <= 0 || a.cap <= 0) return fail("greedy_decode: max_symbols / capacity must be positive");
  if (!a.enc_proj || !a.pred_proj || !a.out_w || !a.out_b || !a.tokens || !a.counts) return fail("greedy_decode: null operand");
  const bool in_lds = (long long)a.V * a.dj <= DEC_W_LDS_FLOATS;
  const size_t lds = (size_t)(DEC_MAX_DJ + DEC_MAX_V + (in_lds ? a.V * a.dj : 0)) * sizeof(float);
  auto k = in_lds ? greedy_decode_kernel<true> : greedy_decode_kernel<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail("greedy_decode: cannot reserve %zu bytes of LDS", lds);
  hipLaunchKernelGGL(k, dim3(a.B), dim3(DEC_THREADS), lds, stream, a.enc_proj, a.ldj, a.pred_proj, a.out_w, a.out_b, a.T,
                     a.dj, a.V, a.blank, a.max_symbols, a.clamp, a.tokens, a.cap, a.counts);
  return check_launch("greedy_decode");
}

}  // namespace fcenc
