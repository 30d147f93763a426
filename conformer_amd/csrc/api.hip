// C ABI of libfcenc_hip.so (declared in include/fcenc.h) and the host-side schedule of one
// encoder forward: which kernel runs on which buffer, in the order of
// FastConformerEncoder.forward (model.py:835-894) and FastConformerBlock.forward (:662-679).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "../../include/fcenc.h"
#include "common.h"
#include "kernels.h"

namespace fcenc {

static thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

hipError_t reserve_lds(const void* kernel, int bytes) {
  struct Entry { const void* kernel; int device; int bytes; };
  static std::mutex mu;
  static std::vector<Entry> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (Entry& it : done)
    if (it.kernel == kernel && it.device == dev) {
      if (it.bytes >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) it.bytes = bytes;
      return e;
    }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.push_back(Entry{kernel, dev, bytes});
  return e;
}

static inline int ceil_half3(int n) {
  for (int i = 0; i < 3; ++i) n = (n + 1) / 2;
  return n;
}
static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Plan {
  int B, T, F, C, d, H, HD, ff, L, es;
  int T1, F1, T2, F2, T3, F3, Tp, M;
  size_t off_stats, off_klen, off_c1, off_c2, off_c3, off_x, off_x2, off_a, off_h, off_xc, off_qkv, off_o, off_g, off_dw, total;
  static constexpr int NSPLIT = 16;
};

static int make_plan(const fcenc_encoder* e, int B, int T, Plan& p) {
  if (!e) return fail("encoder: null descriptor");
  if (e->dtype != FCENC_F32 && e->dtype != FCENC_BF16) return fail("encoder: dtype %d", e->dtype);
  if (B <= 0 || T <= 0) return fail("encoder: empty batch B=%d T=%d", B, T);
  if (e->n_heads <= 0 || e->d_model != e->n_heads * e->head_dim) return fail("encoder: d_model %d != heads %d * head_dim %d", e->d_model, e->n_heads, e->head_dim);
  if (e->head_dim > 64 || e->head_dim % 8) return fail("encoder: head_dim %d unsupported (<= 64, multiple of 8)", e->head_dim);
  if (e->d_model % 128 || e->ff_dim % 128 || e->channels % 128) return fail("encoder: d_model / ff_dim / channels must be multiples of 128");
  if (e->n_layers <= 0 || !e->layers) return fail("encoder: no layers");
  p.B = B; p.T = T; p.F = e->n_mels; p.C = e->channels; p.d = e->d_model; p.H = e->n_heads; p.HD = e->n_heads * 64;
  p.ff = e->ff_dim; p.L = e->n_layers; p.es = e->dtype == FCENC_BF16 ? 2 : 4;
  p.T1 = (T + 1) / 2; p.T2 = (p.T1 + 1) / 2; p.T3 = (p.T2 + 1) / 2;
  p.F1 = (p.F + 1) / 2; p.F2 = (p.F1 + 1) / 2; p.F3 = (p.F2 + 1) / 2;
  p.Tp = (p.T3 + 63) / 64 * 64;
  if (((long long)T * p.F) % 4) return fail("encoder: T * n_mels must be a multiple of 4");
  if ((long long)B * p.Tp >= (1ll << 31) / 8) return fail("encoder: batch too large");
  p.M = B * p.Tp;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
  p.off_stats = take((size_t)B * Plan::NSPLIT * 2 * sizeof(double));
  p.off_klen = take((size_t)B * sizeof(int32_t));
  // with conv1 produced inside conv2 its activation (the largest buffer by far: 1.3 GB at 64 x 20 s) does not exist
  const bool fused12 = e->conv12_a1 && e->conv12_w2 && conv12_supported(e->dtype, p.F, p.C);
  p.off_c1 = take(fused12 ? 0 : (size_t)B * p.T1 * p.F1 * p.C * p.es);
  p.off_c2 = take((size_t)B * p.T2 * p.F2 * p.C * p.es);
  p.off_c3 = take((size_t)B * p.Tp * p.F3 * p.C * p.es);
  p.off_x = take((size_t)p.M * p.d * 4);
  p.off_x2 = take((size_t)p.M * p.d * 4);
  p.off_a = take((size_t)p.M * p.d * p.es);
  p.off_h = take((size_t)p.M * p.ff * p.es);
  p.off_xc = take((size_t)p.M * p.d * p.es);
  p.off_qkv = take((size_t)p.M * 3 * p.HD * p.es);
  p.off_o = take((size_t)p.M * p.HD * p.es);
  p.off_g = take((size_t)p.M * p.d * p.es);
  p.off_dw = take((size_t)p.M * p.d * p.es);
  p.total = o;
  return 0;
}

// the streaming step packs tokens without padding frames: Tp = T3, M = B * T3
static int make_stream_plan(const fcenc_encoder* e, int B, int T, Plan& p) {
  if (make_plan(e, B, T, p)) return -1;
  p.Tp = p.T3;
  p.M = B * p.T3;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align256(bytes); return at; };
  p.off_stats = take((size_t)B * Plan::NSPLIT * 2 * sizeof(double));
  p.off_klen = 0;
  p.off_c1 = take((size_t)B * p.T1 * p.F1 * p.C * p.es);
  p.off_c2 = take((size_t)B * p.T2 * p.F2 * p.C * p.es);
  p.off_c3 = take((size_t)p.M * p.F3 * p.C * p.es);
  p.off_x = take((size_t)p.M * p.d * 4);
  p.off_x2 = 0;
  p.off_a = take((size_t)p.M * p.d * p.es);
  p.off_h = take((size_t)p.M * p.ff * 4);        // hidden activation, or the fp32 partial products of the split FFN
  p.off_xc = take((size_t)p.M * p.d * p.es);
  p.off_qkv = take((size_t)p.M * 3 * p.HD * p.es);
  p.off_o = take((size_t)p.M * p.HD * p.es);
  p.off_g = take((size_t)p.M * p.d * p.es);
  p.off_dw = take((size_t)p.M * p.d * p.es);
  p.total = o;
  return 0;
}

}  // namespace fcenc

using namespace fcenc;

static const int kMaxLanes = 8;      // lanes (slices of a batch on their own HIP streams) a call can be split into

extern "C" {

int fcenc_version(void) { return 100; }
const char* fcenc_last_error(void) { return g_err; }

int fcenc_out_frames(int T) { return ceil_half3(T); }
int fcenc_padded_frames(int T) { return (ceil_half3(T) + 63) / 64 * 64; }

int fcenc_gemm(int dtype, int epilogue, const void* A, const void* W, const float* bias, void* out, float* X,
               void* Xc, int M, int N, int K, int lda, int ldo, float alpha, float beta, void* stream) {
  GemmArgs a;
  a.A = A; a.W = W; a.bias = bias; a.out = out; a.X = X; a.Xc = Xc;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldo = ldo; a.alpha = alpha; a.beta = beta;
  return launch_gemm(dtype, epilogue, a, (hipStream_t)stream);
}

int fcenc_conv3x3s2(int dtype, const void* in, const void* W, const float* bias, void* out, int B, int Hin, int Win,
                    int Cin, int Cout, int HoutPad, void* stream) {
  ConvArgs a;
  a.in = in; a.W = W; a.bias = bias; a.out = out; a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout;
  a.Hout = (Hin + 1) / 2; a.Wout = (Win + 1) / 2; a.HoutPad = HoutPad;
  if (!in || !W || !out) return fail("conv: null operand");
  return launch_conv_igemm(dtype, a, (hipStream_t)stream);
}

int fcenc_input_stats(const float* mel, double* partial, int B, int T, int F, int nsplit, void* stream) {
  if (!mel || !partial) return fail("input_stats: null operand");
  return launch_input_stats(mel, partial, B, T, F, nsplit, (hipStream_t)stream);
}

int fcenc_conv1(int dtype, const float* mel, const double* partial, int nsplit, const float* w9, const float* bias,
                void* out, int B, int T, int F, int C, void* stream) {
  if (!mel || !partial || !w9 || !bias || !out) return fail("conv1: null operand");
  return launch_conv1(dtype, mel, partial, nsplit, w9, bias, out, B, T, F, C, (hipStream_t)stream);
}

int fcenc_conv12_supported(int dtype, int n_mels, int channels) { return conv12_supported(dtype, n_mels, channels) ? 1 : 0; }
long long fcenc_conv12_packed_bytes(int which) { return (long long)(which == 0 ? conv12_a1_bytes() : conv12_w2_bytes()); }
int fcenc_conv12_pack(const float* w9, const float* b1, const void* W2, void* A1p, void* W2p, int n_mels, int channels,
                      void* stream) {
  return launch_conv12_pack(w9, b1, W2, A1p, W2p, n_mels, channels, (hipStream_t)stream);
}
int fcenc_lognorm_conv12(const float* mel, const double* partial, int nsplit, const void* A1p, const void* W2p,
                         const float* bias2, void* out, int B, int T, int n_mels, int channels, void* stream) {
  return launch_conv12(mel, partial, nsplit, A1p, W2p, bias2, out, B, T, n_mels, channels, (hipStream_t)stream);
}

int fcenc_conv3_patch_supported(int dtype, int Win, int Cin, int Cout) {
  ConvArgs a;
  a.Win = Win; a.Wout = (Win + 1) / 2; a.Cin = Cin; a.Cout = Cout;
  return conv3p_supported(dtype, a) ? 1 : 0;
}
long long fcenc_conv3_patch_packed_bytes(void) { return (long long)conv3p_packed_bytes(); }
int fcenc_conv3_patch_pack(const void* W, void* Wp, int channels, void* stream) {
  return launch_conv3p_pack(W, Wp, channels, (hipStream_t)stream);
}
int fcenc_conv3_patch(const void* in, const void* Wp, const float* bias, void* out, int B, int Hin, int Win, int channels,
                      int HoutPad, void* stream) {
  ConvArgs a;
  a.in = in; a.bias = bias; a.out = out; a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = channels; a.Cout = channels;
  a.Hout = (Hin + 1) / 2; a.Wout = (Win + 1) / 2; a.HoutPad = HoutPad;
  return launch_conv3p(a, Wp, (hipStream_t)stream);
}

int fcenc_lognorm_conv1_static(int dtype, const float* mel, const float* w9, const float* bias, void* out, int B, int T,
                               int F, int C, void* stream) {
  if (!mel || !w9 || !bias || !out) return fail("lognorm_conv1_static: null operand");
  return launch_lognorm_conv1_static(dtype, mel, w9, bias, out, B, T, F, C, (hipStream_t)stream);
}

int fcenc_layernorm(int dtype, const float* X, const float* g1, const float* b1, const float* g2, const float* b2,
                    float* Xout, void* Aout, float* compact, int M, int d, int Tp, int Tvalid, void* stream) {
  LayerNormArgs a;
  a.X = X; a.g1 = g1; a.b1 = b1; a.g2 = g2; a.b2 = b2; a.Xout = Xout; a.Aout = Aout; a.compact = compact;
  a.M = M; a.d = d; a.Tp = Tp; a.Tvalid = Tvalid;
  return launch_layernorm(dtype, a, (hipStream_t)stream);
}

int fcenc_layernorm_bwd(const float* X, const float* dY, const float* g, float* dX, float* partial, int nblocks, int M,
                        int d, void* stream) {
  return launch_layernorm_bwd(X, dY, g, dX, partial, nblocks, M, d, (hipStream_t)stream);
}

int fcenc_glu_dwconv_fwd(const float* Y, const float* w9, const float* bias, float* G, float* Z, int B, int T, int d,
                         void* stream) {
  return launch_glu_dwconv_fwd(Y, w9, bias, G, Z, B, T, d, (hipStream_t)stream);
}

int fcenc_glu_dwconv_partial_blocks(int B, int T) { return glu_dwconv_partial_blocks(B, T); }

int fcenc_glu_dwconv_bwd(const float* Y, const float* G, const float* dZ, const float* w9, float* dY, float* partial,
                         int B, int T, int d, void* stream) {
  return launch_glu_dwconv_bwd(Y, G, dZ, w9, dY, partial, B, T, d, (hipStream_t)stream);
}

long long fcenc_linear_bwd_workspace_bytes(int dtype, int M, int N, int K) {
  return (long long)linear_bwd_workspace_bytes(dtype, M, N, K);
}

int fcenc_linear_bwd(int dtype, const float* X, const float* dY, const float* W, float* dX, float* dW, float* db,
                     void* workspace, int M, int N, int K, void* stream) {
  return launch_linear_bwd(dtype, X, dY, W, dX, dW, db, workspace, M, N, K, (hipStream_t)stream);
}

int fcenc_bn_partial_blocks(int M) { return bn_partial_blocks(M); }

int fcenc_bn_silu_train_fwd(const float* Z, const float* g, const float* b, float* running_mean, float* running_var,
                            float momentum, float eps, float* Y, float* stat, float* partial, int M, int d, void* stream) {
  return launch_bn_silu_train_fwd(Z, g, b, running_mean, running_var, momentum, eps, Y, stat, partial, M, d,
                                  (hipStream_t)stream);
}

int fcenc_bn_silu_train_bwd(const float* Z, const float* dY, const float* g, const float* b, const float* stat, float* dZ,
                            float* dg, float* db, float* partial, int M, int d, void* stream) {
  return launch_bn_silu_train_bwd(Z, dY, g, b, stat, dZ, dg, db, partial, M, d, (hipStream_t)stream);
}

int fcenc_ffn_fused_supported(int dtype, int d, int ff) { return ffn_fused_supported(dtype, d, ff) ? 1 : 0; }

int fcenc_ffn_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, void* stream) {
  return launch_ffn_pack(W1, W2, W1p, W2p, d, ff, (hipStream_t)stream);
}

int fcenc_ffn_qkv_fused_supported(int dtype, int d, int ff, int heads) {
  return ffn_qkv_fused_supported(dtype, d, ff, heads) ? 1 : 0;
}

int fcenc_qkv_pack(const void* W, void* Wp, int rows, int d, void* stream) {
  return launch_qkv_pack(W, Wp, rows, d, (hipStream_t)stream);
}

int fcenc_rows_pack(const void* W, void* Wp, int rows, int d, void* stream) {
  return launch_rows_pack(W, Wp, rows, d, (hipStream_t)stream);
}

int fcenc_ffn_fused(float* X, const float* ln_g, const float* ln_b, const void* W1, const float* b1, const void* W2,
                    const float* b2, const float* post_g, const float* post_b, void* Xc, float* compact,
                    const void* Wq, const float* bq, void* Q, int nq, int M, int d, int ff, int Tp, int Tvalid,
                    float alpha, void* stream) {
  FfnArgs a;
  a.X = X; a.ln_g = ln_g; a.ln_b = ln_b; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.post_g = post_g;
  a.post_b = post_b; a.Xc = Xc; a.compact = compact; a.M = M; a.d = d; a.ff = ff; a.Tp = Tp; a.Tvalid = Tvalid;
  a.alpha = alpha; a.Wq = Wq; a.bq = bq; a.Q = Q; a.nq = nq;
  return launch_ffn_fused(a, (hipStream_t)stream);
}

int fcenc_convmod_fused_supported(int dtype, int d, int kernel_size) {
  return convmod_fused_supported(dtype, d, kernel_size) ? 1 : 0;
}

int fcenc_convmod_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, void* stream) {
  return launch_convmod_pack(W1, W2, W1p, W2p, d, (hipStream_t)stream);
}

int fcenc_convmod_oproj_supported(int dtype, int d, int kernel_size, int heads) {
  return convmod_oproj_supported(dtype, d, kernel_size, heads) ? 1 : 0;
}

int fcenc_convmod_fused(const float* X, float* Xout, const float* ln_g, const float* ln_b, const void* W1,
                        const float* b1, const float* dw_w9, const float* dw_b, const void* W2, const float* b2,
                        const void* O, const void* Wo, const float* bo, int M, int d, int Tp, void* stream) {
  ConvModArgs a;
  a.X = X; a.Xout = Xout; a.ln_g = ln_g; a.ln_b = ln_b; a.W1 = W1; a.b1 = b1; a.dw_w9 = dw_w9; a.dw_b = dw_b;
  a.W2 = W2; a.b2 = b2; a.O = O; a.Wo = Wo; a.bo = bo; a.M = M; a.d = d; a.Tp = Tp;
  return launch_convmod_fused(a, (hipStream_t)stream);
}

int fcenc_block_tail(const fcenc_layer* w, const float* X, float* Xout, const void* O, float* compact, int M, int d,
                     int ff, int heads, int Tp, int Tvalid, void* stream) {
  if (!w) return fail("block_tail: null layer");
  if (!convmod_oproj_supported(DT_BF16, d, 9, heads) || !ffn_fused_supported(DT_BF16, d, ff))
    return fail("block_tail: unsupported shape d=%d ff=%d heads=%d", d, ff, heads);
  FfnArgs f;
  f.ln_g = (const float*)w->ff2_ln_g; f.ln_b = (const float*)w->ff2_ln_b; f.W1 = w->ff2_w1; f.b1 = (const float*)w->ff2_b1;
  f.W2 = w->ff2_w2; f.b2 = (const float*)w->ff2_b2; f.post_g = (const float*)w->ln_g; f.post_b = (const float*)w->ln_b;
  f.M = M; f.d = d; f.ff = ff; f.alpha = 0.5f; f.Tp = Tp; f.Tvalid = Tvalid; f.compact = compact;
  ConvModArgs c;
  c.X = X; c.Xout = Xout; c.ln_g = (const float*)w->conv_ln_g; c.ln_b = (const float*)w->conv_ln_b; c.W1 = w->pw1_w;
  c.b1 = (const float*)w->pw1_b; c.dw_w9 = (const float*)w->dw_w9; c.dw_b = (const float*)w->dw_b; c.W2 = w->pw2_w;
  c.b2 = (const float*)w->pw2_b; c.O = O; c.Wo = w->out_w; c.bo = (const float*)w->out_b; c.M = M; c.d = d; c.Tp = Tp;
  c.ffn = &f;
  return launch_convmod_fused(c, (hipStream_t)stream);
}

int fcenc_dwconv(int dtype, const void* in, const float* w9, const float* bias, void* out, int B, int Tp, int d,
                 void* stream) {
  if (!in || !w9 || !bias || !out) return fail("dwconv: null operand");
  return launch_dwconv(dtype, in, w9, bias, out, B, Tp, d, (hipStream_t)stream);
}

int fcenc_attention(int dtype, const void* qkv, const void* pos, const float* bias_u, const float* bias_v,
                    const int32_t* klen, void* out, int B, int H, int Tp, int Tvalid, int window, float scale,
                    void* stream) {
  AttnArgs a;
  a.qkv = qkv; a.pos = pos; a.bias_u = bias_u; a.bias_v = bias_v; a.klen = klen; a.out = out;
  a.B = B; a.H = H; a.Tp = Tp; a.Tvalid = Tvalid; a.window = window; a.scale = scale;
  return launch_attention(dtype, a, (hipStream_t)stream);
}

int fcenc_lengths(const int64_t* lengths, int64_t* out_len, uint8_t* mask, int32_t* klen, int B, int Tvalid,
                  void* stream) {
  return launch_lengths(lengths, out_len, mask, klen, B, Tvalid, (hipStream_t)stream);
}

size_t fcenc_encoder_workspace_bytes(const fcenc_encoder* enc, int B, int T) {
  Plan p;
  if (make_plan(enc, B, T, p)) return 0;
  return p.total;
}

int fcenc_encoder_forward_lanes(const fcenc_encoder* e, const float* mel, const int64_t* lengths, int B, int T,
                                float* out, int64_t* out_len, uint8_t* mask, void* workspace, size_t workspace_bytes,
                                void* stream_, const fcenc_stream_lanes* side) {
  Plan p;
  const int n_lanes = 1 + (side ? side->n : 0);
  if (side && (side->n < 0 || side->n >= kMaxLanes)) return fail("encoder: %d side streams (0..%d)", side->n, kMaxLanes - 1);
  if (n_lanes > B) return fail("encoder: %d utterances cannot be split over %d lanes", B, n_lanes);
  if (n_lanes > 1) {
    if (!side->events[0]) return fail("encoder: no fork event");
    for (int g = 1; g < n_lanes; ++g)
      if (!side->streams[g - 1] || !side->events[g]) return fail("encoder: lane %d has no stream / event", g);
  }
  if (make_plan(e, B, T, p)) return -1;
  if (!mel || !out || !out_len || !workspace) return fail("encoder: null buffer");
  if (workspace_bytes < p.total) return fail("encoder: workspace %zu < %zu bytes", workspace_bytes, p.total);
  if ((uintptr_t)workspace & 255) return fail("encoder: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream_;
  float* const out_all = out;
  const bool first_ff1_done = false;
  char* ws = (char*)workspace;
  const int dt = e->dtype;
  double* stats = (double*)(ws + p.off_stats);
  int32_t* klen = (int32_t*)(ws + p.off_klen);
  void *c1 = ws + p.off_c1, *c2 = ws + p.off_c2, *c3 = ws + p.off_c3;
  float* X = (float*)(ws + p.off_x);      // residual stream; swaps with X2 at every fused conv module
  float* X2 = (float*)(ws + p.off_x2);
  void *A = ws + p.off_a, *Hb = ws + p.off_h, *Xc = ws + p.off_xc, *QKV = ws + p.off_qkv, *O = ws + p.off_o,
       *G = ws + p.off_g, *DW = ws + p.off_dw;

#define FCENC_TRY(x) do { if ((x) != 0) return -1; } while (0)
  // lengths / padding mask (model.py:863-876)
  // The same launch zeroes frames T3..Tp-1 of conv3's output: they are layout padding and must be
  // finite, so that no NaN/Inf garbage can reach a real frame through 0 * NaN in the attention's P.V
  {
    const size_t rowb = (size_t)p.F3 * p.C * p.es;
    FCENC_TRY(launch_lengths(lengths, out_len, lengths ? mask : nullptr, klen, B, p.T3, s,
                             p.Tp > p.T3 ? (char*)c3 + (size_t)p.T3 * rowb : nullptr, (size_t)p.Tp * rowb,
                             (size_t)(p.Tp - p.T3) * rowb));
  }

  // input standardisation + subsampling convs (model.py:856-857, :741-743)
  FCENC_TRY(launch_input_stats(mel, stats, B, T, p.F, Plan::NSPLIT, s));
  ConvArgs cv;
  cv.in = c1; cv.W = e->conv2_w; cv.bias = e->conv2_b; cv.out = c2; cv.B = B; cv.Hin = p.T1; cv.Win = p.F1;
  cv.Cin = p.C; cv.Cout = p.C; cv.Hout = p.T2; cv.Wout = p.F2; cv.HoutPad = p.T2;
  if (e->conv12_a1 && e->conv12_w2 && conv12_supported(dt, p.F, p.C)) {
    // conv1 produced inside conv2: one launch, no conv1 activation in memory (the events bracket that launch)
    if (e->ev_begin && hipEventRecord((hipEvent_t)e->ev_begin, s) != hipSuccess) return fail("encoder: event record");
    FCENC_TRY(launch_conv12(mel, stats, Plan::NSPLIT, e->conv12_a1, e->conv12_w2, e->conv2_b, c2, B, T, p.F, p.C, s));
    if (e->ev_end && hipEventRecord((hipEvent_t)e->ev_end, s) != hipSuccess) return fail("encoder: event record");
  } else {
    FCENC_TRY(launch_conv1(dt, mel, stats, Plan::NSPLIT, e->conv1_w9, e->conv1_b, c1, B, T, p.F, p.C, s));
    if (e->ev_begin && hipEventRecord((hipEvent_t)e->ev_begin, s) != hipSuccess) return fail("encoder: event record");
    FCENC_TRY(launch_conv_igemm(dt, cv, s));
    if (e->ev_end && hipEventRecord((hipEvent_t)e->ev_end, s) != hipSuccess) return fail("encoder: event record");
  }
  cv.in = c2; cv.W = e->conv3_w; cv.bias = e->conv3_b; cv.out = c3; cv.Hin = p.T2; cv.Win = p.F2;
  cv.Hout = p.T3; cv.Wout = p.F3; cv.HoutPad = p.Tp;
  if (e->conv3_wp && conv3p_supported(dt, cv)) FCENC_TRY(launch_conv3p(cv, e->conv3_wp, s));
  else FCENC_TRY(launch_conv_igemm(dt, cv, s));

  // subsampling Linear, times sqrt(d) (model.py:751, :172)
  GemmArgs g;
  g.A = c3; g.W = e->lin_w; g.bias = e->lin_b; g.X = X; g.Xc = nullptr; g.M = p.M; g.N = p.d; g.K = p.F3 * p.C;
  g.lda = g.K; g.alpha = sqrtf((float)p.d); g.beta = 0.f;
  FCENC_TRY(launch_gemm(dt, EPI_RESID, g, s));

  // the fused feed-forward kernel owns its LayerNorms; the unfused schedule starts with ff1's
  const bool fused_ffn = ffn_fused_supported(dt, p.d, p.ff);
  const bool fused_conv = convmod_fused_supported(dt, p.d, 9);
  const bool fused_qkv = ffn_qkv_fused_supported(dt, p.d, p.ff, p.H);
  const bool fused_oproj = convmod_oproj_supported(dt, p.d, 9, p.H);
  if (!fused_ffn) {
    LayerNormArgs ln;
    ln.X = X; ln.M = p.M; ln.d = p.d; ln.Tp = p.Tp; ln.Tvalid = p.T3;
    ln.g1 = e->layers[0].ff1_ln_g; ln.b1 = e->layers[0].ff1_ln_b; ln.Aout = A;
    FCENC_TRY(launch_layernorm(dt, ln, s));
  }

  const float scale = 1.0f / sqrtf((float)e->head_dim);
  const int window = (e->window > 0 && p.T3 > e->window) ? e->window : 0;   // model.py:406

  // The blocks of utterances [b0, b0 + nb) on stream `s`.  Utterances are independent from here on and every buffer
  // is row-major by token, so a contiguous slice of the batch is a contiguous slice of every buffer.
  float* const X_all = X;
  float* const X2_all = X2;
  auto run_layers = [&](const int b0, const int nb, hipStream_t s) -> int {
    const size_t m0 = (size_t)b0 * p.Tp;
    const int M = nb * p.Tp, B = nb;
    float* X = X_all + m0 * p.d;
    float* X2 = X2_all + m0 * p.d;
    void* A = (char*)(ws + p.off_a) + m0 * p.d * p.es;
    void* Hb = (char*)(ws + p.off_h) + m0 * p.ff * p.es;
    void* Xc = (char*)(ws + p.off_xc) + m0 * p.d * p.es;
    void* QKV = (char*)(ws + p.off_qkv) + m0 * 3 * p.HD * p.es;
    void* O = (char*)(ws + p.off_o) + m0 * p.HD * p.es;
    void* G = (char*)(ws + p.off_g) + m0 * p.d * p.es;
    void* DW = (char*)(ws + p.off_dw) + m0 * p.d * p.es;
    const int32_t* klen = (const int32_t*)(ws + p.off_klen) + b0;
    float* out = out_all + (size_t)b0 * p.T3 * p.d;
    bool ff1_done = first_ff1_done;
    for (int l = 0; l < p.L; ++l) {
      const fcenc_layer& w = e->layers[l];
      if (!w.pos) return fail("encoder: layer %d has no position table", l);
      auto gemm = [&](int epi, const void* a, int lda, const void* wt, const float* bias, int N, int K, void* o, int ldo,
                      float alpha, void* xc) {
        GemmArgs q;
        q.A = a; q.W = wt; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.out = o; q.ldo = ldo;
        q.X = X; q.Xc = xc; q.alpha = alpha; q.beta = 1.f;
        return launch_gemm(dt, epi, q, s);
      };
      // x += 0.5 * ff1(x)   (model.py:663); Xc = x in the compute dtype for the QKV projection
      if (ff1_done) {
        ff1_done = false;       // the previous block's last kernel has already done this and the projection
      } else if (fused_ffn) {
        FfnArgs f;
        f.X = X; f.ln_g = w.ff1_ln_g; f.ln_b = w.ff1_ln_b; f.W1 = w.ff1_w1; f.b1 = w.ff1_b1; f.W2 = w.ff1_w2;
        f.b2 = w.ff1_b2; f.M = M; f.d = p.d; f.ff = p.ff; f.alpha = 0.5f;
        if (fused_qkv) {
          f.Wq = w.qkv_w; f.bq = w.qkv_b; f.Q = QKV; f.nq = 3 * p.HD;     // the same kernel projects Q | K | V
        } else {
          f.Xc = Xc;
        }
        FCENC_TRY(launch_ffn_fused(f, s));
      } else {
        FCENC_TRY(gemm(EPI_STORE_SILU, A, p.d, w.ff1_w1, w.ff1_b1, p.ff, p.d, Hb, p.ff, 1.f, nullptr));
        FCENC_TRY(gemm(EPI_RESID, Hb, p.ff, w.ff1_w2, w.ff1_b2, p.d, p.ff, nullptr, 0, 0.5f, Xc));
      }
      // x += attention(x)   (model.py:666-667; no LayerNorm in front)
      if (!fused_qkv) FCENC_TRY(gemm(EPI_STORE, Xc, p.d, w.qkv_w, w.qkv_b, 3 * p.HD, p.d, QKV, 3 * p.HD, 1.f, nullptr));
      AttnArgs at;
      at.qkv = QKV; at.pos = w.pos; at.bias_u = w.bias_u; at.bias_v = w.bias_v; at.klen = klen; at.out = O;
      at.B = B; at.H = p.H; at.Tp = p.Tp; at.Tvalid = p.T3; at.window = window; at.scale = scale;
      FCENC_TRY(launch_attention(dt, at, s));
      const bool timed = l == p.L / 2 && b0 == 0 && e->ev_block_begin && e->ev_block_end;
      if (timed && hipEventRecord((hipEvent_t)e->ev_block_begin, s) != hipSuccess) return fail("encoder: event record");
      if (!fused_oproj) FCENC_TRY(gemm(EPI_RESID, O, p.HD, w.out_w, w.out_b, p.d, p.HD, nullptr, 0, 1.f, nullptr));
      // x += conv(x)        (model.py:670-671)
      LayerNormArgs l1;
      l1.X = X; l1.M = M; l1.d = p.d; l1.g1 = w.conv_ln_g; l1.b1 = w.conv_ln_b; l1.Aout = A;
      if (fused_conv) {
        ConvModArgs c;
        c.X = X; c.Xout = X2; c.ln_g = w.conv_ln_g; c.ln_b = w.conv_ln_b; c.W1 = w.pw1_w; c.b1 = w.pw1_b;
        c.dw_w9 = w.dw_w9; c.dw_b = w.dw_b; c.W2 = w.pw2_w; c.b2 = w.pw2_b; c.M = M; c.d = p.d; c.Tp = p.Tp;
        FfnArgs f2, fn;
        if (fused_oproj) {
          c.O = O; c.Wo = w.out_w; c.bo = w.out_b;      // the same kernel first adds out_proj(attention)
          if (fused_ffn) {
            // ... and ends with x += 0.5 * ff2(x), x = LayerNorm(x) (model.py:674, :677)
            f2.ln_g = w.ff2_ln_g; f2.ln_b = w.ff2_ln_b; f2.W1 = w.ff2_w1; f2.b1 = w.ff2_b1; f2.W2 = w.ff2_w2;
            f2.b2 = w.ff2_b2; f2.post_g = w.ln_g; f2.post_b = w.ln_b; f2.M = M; f2.d = p.d; f2.ff = p.ff;
            f2.alpha = 0.5f; f2.Tp = p.Tp; f2.Tvalid = p.T3;
            if (l + 1 == p.L) f2.compact = out;
            c.ffn = &f2;
            if (l + 1 < p.L && fused_qkv) {
              // ... and goes straight on with the next block's x += 0.5 * ff1(x) and q | k | v projection
              const fcenc_layer& n = e->layers[l + 1];
              fn.ln_g = n.ff1_ln_g; fn.ln_b = n.ff1_ln_b; fn.W1 = n.ff1_w1; fn.b1 = n.ff1_b1; fn.W2 = n.ff1_w2;
              fn.b2 = n.ff1_b2; fn.M = M; fn.d = p.d; fn.ff = p.ff; fn.alpha = 0.5f;
              fn.Wq = n.qkv_w; fn.bq = n.qkv_b; fn.Q = QKV; fn.nq = 3 * p.HD;
              c.next = &fn;
              ff1_done = true;
            }
          }
        }
        FCENC_TRY(launch_convmod_fused(c, s));
        float* t = X; X = X2; X2 = t;
        l1.X = X;
        if (c.ffn) {
          if (timed && hipEventRecord((hipEvent_t)e->ev_block_end, s) != hipSuccess) return fail("encoder: event record");
          continue;
        }
      } else {
        FCENC_TRY(launch_layernorm(dt, l1, s));
        FCENC_TRY(gemm(EPI_GLU, A, p.d, w.pw1_w, w.pw1_b, 2 * p.d, p.d, G, p.d, 1.f, nullptr));
        FCENC_TRY(launch_dwconv(dt, G, w.dw_w9, w.dw_b, DW, B, p.Tp, p.d, s));
        FCENC_TRY(gemm(EPI_RESID, DW, p.d, w.pw2_w, w.pw2_b, p.d, p.d, nullptr, 0, 1.f, nullptr));
      }
      // x += 0.5 * ff2(x), x = LayerNorm(x)   (model.py:674, :677)
      if (fused_ffn) {
        FfnArgs f;
        f.X = X; f.ln_g = w.ff2_ln_g; f.ln_b = w.ff2_ln_b; f.W1 = w.ff2_w1; f.b1 = w.ff2_b1; f.W2 = w.ff2_w2;
        f.b2 = w.ff2_b2; f.post_g = w.ln_g; f.post_b = w.ln_b; f.M = M; f.d = p.d; f.ff = p.ff; f.alpha = 0.5f;
        f.Tp = p.Tp; f.Tvalid = p.T3;
        if (l + 1 == p.L) f.compact = out;
        FCENC_TRY(launch_ffn_fused(f, s));
        if (timed && hipEventRecord((hipEvent_t)e->ev_block_end, s) != hipSuccess) return fail("encoder: event record");
        continue;
      }
      l1.g1 = w.ff2_ln_g; l1.b1 = w.ff2_ln_b;
      FCENC_TRY(launch_layernorm(dt, l1, s));
      FCENC_TRY(gemm(EPI_STORE_SILU, A, p.d, w.ff2_w1, w.ff2_b1, p.ff, p.d, Hb, p.ff, 1.f, nullptr));
      FCENC_TRY(gemm(EPI_RESID, Hb, p.ff, w.ff2_w2, w.ff2_b2, p.d, p.ff, nullptr, 0, 0.5f, nullptr));
      // x = LayerNorm(x)    (model.py:677), chained with the next block's ff1 LayerNorm
      LayerNormArgs l2;
      l2.X = X; l2.M = M; l2.d = p.d; l2.Tp = p.Tp; l2.Tvalid = p.T3; l2.g1 = w.ln_g; l2.b1 = w.ln_b;
      if (l + 1 < p.L) {
        l2.Xout = X; l2.g2 = e->layers[l + 1].ff1_ln_g; l2.b2 = e->layers[l + 1].ff1_ln_b; l2.Aout = A;
      } else {
        l2.compact = out;
      }
      FCENC_TRY(launch_layernorm(dt, l2, s));
      if (timed && hipEventRecord((hipEvent_t)e->ev_block_end, s) != hipSuccess) return fail("encoder: event record");
    }
    return 0;
  };

  // fork: the side streams start behind the front-end queued on `stream`; lanes are queued whole, one after the
  // other (a lane's launches are dependent, so nothing is gained by interleaving them on the host)
  if (n_lanes > 1) {
    if (hipEventRecord((hipEvent_t)side->events[0], s) != hipSuccess) return fail("encoder: event record");
    for (int g = 1; g < n_lanes; ++g)
      if (hipStreamWaitEvent((hipStream_t)side->streams[g - 1], (hipEvent_t)side->events[0], 0) != hipSuccess)
        return fail("encoder: stream wait");
  }
  int rc = 0;
  for (int g = 0; g < n_lanes && rc == 0; ++g) {
    const int b0 = (int)((long long)B * g / n_lanes), b1 = (int)((long long)B * (g + 1) / n_lanes);
    rc = run_layers(b0, b1 - b0, g == 0 ? s : (hipStream_t)side->streams[g - 1]);
  }
  // join, also after a failed launch: `stream` must not run ahead of work already queued on the side streams
  for (int g = 1; g < n_lanes; ++g) {
    if (hipEventRecord((hipEvent_t)side->events[g], (hipStream_t)side->streams[g - 1]) != hipSuccess ||
        hipStreamWaitEvent(s, (hipEvent_t)side->events[g], 0) != hipSuccess)
      return fail("encoder: join failed");
  }
  if (rc != 0) return -1;

#undef FCENC_TRY
  return 0;
}

int fcenc_encoder_forward(const fcenc_encoder* e, const float* mel, const int64_t* lengths, int B, int T, float* out,
                          int64_t* out_len, uint8_t* mask, void* workspace, size_t workspace_bytes, void* stream_) {
  return fcenc_encoder_forward_lanes(e, mel, lengths, B, T, out, out_len, mask, workspace, workspace_bytes, stream_,
                                     nullptr);
}

int fcenc_stream_attention(int dtype, const void* qkv, const void* pos, const float* bias_u, const float* bias_v,
                           void* kcache, void* vcache, void* out, int B, int H, int Tq, int cap, long long frames,
                           float scale, void* stream) {
  StreamAttnArgs a;
  a.qkv = qkv; a.pos = pos; a.bias_u = bias_u; a.bias_v = bias_v; a.kcache = kcache; a.vcache = vcache; a.out = out;
  a.B = B; a.H = H; a.Tq = Tq; a.cap = cap; a.frames = frames; a.scale = scale;
  return launch_stream_attention(dtype, a, (hipStream_t)stream);
}

int fcenc_stream_dwconv(int dtype, const void* g, void* cache, const float* w9, const float* bias, void* out, int B,
                        int Tq, int d, void* stream) {
  return launch_stream_dwconv(dtype, g, cache, w9, bias, out, B, Tq, d, (hipStream_t)stream);
}

int fcenc_stream_max_chunk_frames(void) { return stream_max_chunk_frames(); }
int fcenc_stream_max_cache_frames(void) { return stream_max_cache_frames(); }

// the streams [b0, b0 + nb) of a streaming step, on their own HIP stream and workspace slice
struct StreamLane {
  const fcenc_encoder* e;
  const fcenc_stream_state* st;
  Plan p;
  int b0, nb;
  size_t ws_off;          // this lane's slice of the workspace
  hipStream_t s;
  const float* mel;
  float* out;
  double* stats;
  void *c1, *c2, *c3, *A, *Hb, *Xc, *QKV, *O, *G, *DW;
  float* X;
  bool fused_ffn, fused_qkv, split_ffn;

  void bind(char* ws) {
    stats = (double*)(ws + p.off_stats);
    c1 = ws + p.off_c1; c2 = ws + p.off_c2; c3 = ws + p.off_c3;
    X = (float*)(ws + p.off_x);
    A = ws + p.off_a; Hb = ws + p.off_h; Xc = ws + p.off_xc; QKV = ws + p.off_qkv; O = ws + p.off_o;
    G = ws + p.off_g; DW = ws + p.off_dw;
    fused_ffn = ffn_fused_supported(e->dtype, p.d, p.ff);
    fused_qkv = ffn_qkv_fused_supported(e->dtype, p.d, p.ff, p.H);
    // a chunk's few tokens leave most CUs idle at one workgroup per 64 tokens: split the hidden dimension over CUs
    split_ffn = fused_ffn && ffn_split_pays(p.M, p.d, p.ff);
  }
  int gemm(int epi, const void* a, int lda, const void* wt, const float* bias, int N, int K, void* o, int ldo, float alpha,
           void* xc) const {
    GemmArgs q;
    q.A = a; q.W = wt; q.bias = bias; q.M = p.M; q.N = N; q.K = K; q.lda = lda; q.out = o; q.ldo = ldo;
    q.X = X; q.Xc = xc; q.alpha = alpha; q.beta = 1.f;
    return launch_gemm(e->dtype, epi, q, s);
  }
  int layernorm(const float* g1, const float* b1, const float* g2, const float* b2, bool inplace, float* compact) const {
    LayerNormArgs ln;
    ln.X = X; ln.M = p.M; ln.d = p.d; ln.Tp = p.T3; ln.Tvalid = p.T3; ln.g1 = g1; ln.b1 = b1; ln.g2 = g2; ln.b2 = b2;
    ln.Xout = inplace ? X : nullptr; ln.Aout = compact ? nullptr : A; ln.compact = compact;
    return launch_layernorm(e->dtype, ln, s);
  }

#define FCENC_TRY(x) do { if ((x) != 0) return -1; } while (0)
  // per-chunk standardisation and un-cached subsampling (model.py:856-857, :741-751)
  int front() const {
    const int dt = e->dtype;
    FCENC_TRY(launch_input_stats(mel, stats, nb, p.T, p.F, Plan::NSPLIT, s));
    FCENC_TRY(launch_conv1(dt, mel, stats, Plan::NSPLIT, e->conv1_w9, e->conv1_b, c1, nb, p.T, p.F, p.C, s));
    ConvArgs cv;
    cv.in = c1; cv.W = e->conv2_w; cv.bias = e->conv2_b; cv.out = c2; cv.B = nb; cv.Hin = p.T1; cv.Win = p.F1;
    cv.Cin = p.C; cv.Cout = p.C; cv.Hout = p.T2; cv.Wout = p.F2; cv.HoutPad = p.T2;
    FCENC_TRY(launch_conv_igemm(dt, cv, s));
    cv.in = c2; cv.W = e->conv3_w; cv.bias = e->conv3_b; cv.out = c3; cv.Hin = p.T2; cv.Win = p.F2;
    cv.Hout = p.T3; cv.Wout = p.F3; cv.HoutPad = p.T3;
    FCENC_TRY(launch_conv_igemm(dt, cv, s));
    GemmArgs g;
    g.A = c3; g.W = e->lin_w; g.bias = e->lin_b; g.X = X; g.M = p.M; g.N = p.d; g.K = p.F3 * p.C;
    g.lda = g.K; g.alpha = sqrtf((float)p.d); g.beta = 0.f;
    FCENC_TRY(launch_gemm(dt, EPI_RESID, g, s));
    if (!fused_ffn) FCENC_TRY(layernorm(e->layers[0].ff1_ln_g, e->layers[0].ff1_ln_b, nullptr, nullptr, false, nullptr));
    return 0;
  }

  int layer(int l) const {
    const int dt = e->dtype;
    const size_t es = p.es;
    const fcenc_layer& w = e->layers[l];
    const fcenc_stream_layer& c = st->layers[l];
    // x += 0.5 * ff1(x)   (model.py:663), q | k | v of the chunk (:348-350)
    // the split schedule projects with a plain GEMM, which needs the row-major matrix
    const bool split1 = split_ffn && (!fused_qkv || c.qkv_w);
    const bool qkv_in_ffn = fused_qkv && !split1;
    if (fused_ffn) {
      FfnArgs f;
      f.X = X; f.ln_g = w.ff1_ln_g; f.ln_b = w.ff1_ln_b; f.W1 = w.ff1_w1; f.b1 = w.ff1_b1; f.W2 = w.ff1_w2;
      f.b2 = w.ff1_b2; f.M = p.M; f.d = p.d; f.ff = p.ff; f.alpha = 0.5f;
      if (qkv_in_ffn) {
        f.Wq = w.qkv_w; f.bq = w.qkv_b; f.Q = QKV; f.nq = 3 * p.HD;
      } else {
        f.Xc = Xc;
      }
      if (split1) f.partial = (float*)Hb;
      FCENC_TRY(launch_ffn_fused(f, s));
    } else {
      FCENC_TRY(gemm(EPI_STORE_SILU, A, p.d, w.ff1_w1, w.ff1_b1, p.ff, p.d, Hb, p.ff, 1.f, nullptr));
      FCENC_TRY(gemm(EPI_RESID, Hb, p.ff, w.ff1_w2, w.ff1_b2, p.d, p.ff, nullptr, 0, 0.5f, Xc));
    }
    if (!qkv_in_ffn)
      FCENC_TRY(gemm(EPI_STORE, Xc, p.d, fused_qkv ? c.qkv_w : w.qkv_w, w.qkv_b, 3 * p.HD, p.d, QKV, 3 * p.HD, 1.f, nullptr));
    // x += out_proj(attention over cache | chunk)   (model.py:352-431, :666-667)
    const size_t ring = (size_t)p.H * st->cap * 64 * es;       // one stream's K (or V) ring
    StreamAttnArgs at;
    at.qkv = QKV; at.pos = c.pos; at.bias_u = w.bias_u; at.bias_v = w.bias_v;
    at.kcache = (char*)c.kcache + (size_t)b0 * ring; at.vcache = (char*)c.vcache + (size_t)b0 * ring; at.out = O;
    at.B = nb; at.H = p.H; at.Tq = p.T3; at.cap = st->cap; at.frames = st->frames; at.dyn = st->ring;
    at.scale = 1.0f / sqrtf((float)e->head_dim);
    FCENC_TRY(launch_stream_attention(dt, at, s));
    if (stream_convmod_supported(dt, p.d, p.H, p.T3) && convmod_oproj_supported(dt, p.d, 9, p.H)) {
      // out_proj + residual + conv module in one launch (the layer descriptor's matrices are in streaming order)
      StreamConvModArgs cm;
      cm.X = X; cm.O = O; cm.Wo = w.out_w; cm.bo = w.out_b; cm.ln_g = w.conv_ln_g; cm.ln_b = w.conv_ln_b; cm.W1 = w.pw1_w;
      cm.b1 = w.pw1_b; cm.dw_w9 = w.dw_w9; cm.dw_b = w.dw_b; cm.W2 = w.pw2_w; cm.b2 = w.pw2_b;
      cm.cache = (char*)c.conv_cache + (size_t)b0 * 8 * p.d * es; cm.B = nb; cm.Tq = p.T3; cm.d = p.d; cm.heads = p.H;
      FCENC_TRY(launch_stream_convmod(cm, s));
    } else {
    FCENC_TRY(gemm(EPI_RESID, O, p.HD, c.out_w, w.out_b, p.d, p.HD, nullptr, 0, 1.f, nullptr));
    // x += conv(x) with the depthwise conv's left context from the cache   (model.py:516-544, :670-671)
    FCENC_TRY(layernorm(w.conv_ln_g, w.conv_ln_b, nullptr, nullptr, false, nullptr));
    FCENC_TRY(gemm(EPI_GLU, A, p.d, c.pw1_w, w.pw1_b, 2 * p.d, p.d, G, p.d, 1.f, nullptr));
    FCENC_TRY(launch_stream_dwconv(dt, G, (char*)c.conv_cache + (size_t)b0 * 8 * p.d * es, w.dw_w9, w.dw_b, DW, nb, p.T3,
                                   p.d, s));
    FCENC_TRY(gemm(EPI_RESID, DW, p.d, c.pw2_w, w.pw2_b, p.d, p.d, nullptr, 0, 1.f, nullptr));
    }
    // x += 0.5 * ff2(x), x = LayerNorm(x)   (model.py:674, :677)
    if (fused_ffn) {
      FfnArgs f;
      f.X = X; f.ln_g = w.ff2_ln_g; f.ln_b = w.ff2_ln_b; f.W1 = w.ff2_w1; f.b1 = w.ff2_b1; f.W2 = w.ff2_w2;
      f.b2 = w.ff2_b2; f.post_g = w.ln_g; f.post_b = w.ln_b; f.M = p.M; f.d = p.d; f.ff = p.ff; f.alpha = 0.5f;
      f.Tp = p.T3; f.Tvalid = p.T3;
      if (l + 1 == p.L) f.compact = out;
      if (split_ffn) f.partial = (float*)Hb;
      return launch_ffn_fused(f, s);
    }
    FCENC_TRY(layernorm(w.ff2_ln_g, w.ff2_ln_b, nullptr, nullptr, false, nullptr));
    FCENC_TRY(gemm(EPI_STORE_SILU, A, p.d, w.ff2_w1, w.ff2_b1, p.ff, p.d, Hb, p.ff, 1.f, nullptr));
    FCENC_TRY(gemm(EPI_RESID, Hb, p.ff, w.ff2_w2, w.ff2_b2, p.d, p.ff, nullptr, 0, 0.5f, nullptr));
    if (l + 1 < p.L)
      return layernorm(w.ln_g, w.ln_b, e->layers[l + 1].ff1_ln_g, e->layers[l + 1].ff1_ln_b, true, nullptr);
    return layernorm(w.ln_g, w.ln_b, nullptr, nullptr, false, out);
  }
#undef FCENC_TRY
};


// lane g of n takes streams [B g / n, B (g + 1) / n)
static int plan_lanes(const fcenc_encoder* e, int B, int T, int n, StreamLane* lanes, size_t* total) {
  if (n < 1 || n > kMaxLanes) return fail("stream step: %d lanes (1..%d)", n, kMaxLanes);
  if (B < n) return fail("stream step: %d streams cannot be split over %d lanes", B, n);
  size_t o = 0;
  for (int g = 0; g < n; ++g) {
    StreamLane& L = lanes[g];
    L.e = e;
    L.b0 = (int)((long long)B * g / n);
    L.nb = (int)((long long)B * (g + 1) / n) - L.b0;
    if (make_stream_plan(e, L.nb, T, L.p)) return -1;
    L.ws_off = o;           // sizes are multiples of 256, so every slice stays aligned
    o += L.p.total;
  }
  *total = o;
  return 0;
}

size_t fcenc_encoder_stream_workspace_bytes(const fcenc_encoder* enc, int B, int T, int n_lanes) {
  StreamLane lanes[kMaxLanes];
  size_t total = 0;
  if (plan_lanes(enc, B, T, n_lanes, lanes, &total)) return 0;
  return total;
}

int fcenc_encoder_stream_step(const fcenc_encoder* e, fcenc_stream_state* st, const float* mel, int T, float* out,
                              void* workspace, size_t workspace_bytes, void* stream_, const fcenc_stream_lanes* side) {
  if (!st || !st->layers) return fail("stream step: null state");
  const int B = st->batch;
  const int n = 1 + (side ? side->n : 0);
  StreamLane lanes[kMaxLanes];
  size_t total = 0;
  if (side && (side->n < 0 || side->n >= kMaxLanes)) return fail("stream step: %d side streams (0..%d)", side->n, kMaxLanes - 1);
  if (plan_lanes(e, B, T, n, lanes, &total)) return -1;
  const Plan& p0 = lanes[0].p;
  if (!mel || !out || !workspace) return fail("stream step: null buffer");
  if (workspace_bytes < total) return fail("stream step: workspace %zu < %zu bytes", workspace_bytes, total);
  if ((uintptr_t)workspace & 255) return fail("stream step: workspace must be 256-byte aligned");
  if (p0.T3 > stream_max_chunk_frames()) return fail("stream step: %d frames per chunk (at most %d)", p0.T3, stream_max_chunk_frames());
  if (st->cap < p0.T3 || st->cap > stream_max_cache_frames() || st->cap % 8)
    return fail("stream step: cache of %d frames out of range (%d..%d, multiple of 8)", st->cap, p0.T3, stream_max_cache_frames());
  if (st->frames < 0) return fail("stream step: negative frame count");
  for (int l = 0; l < p0.L; ++l) {
    const fcenc_stream_layer& c = st->layers[l];
    if (!c.kcache || !c.vcache || !c.conv_cache || !c.pos || !c.out_w || !c.pw1_w || !c.pw2_w)
      return fail("stream step: layer %d state incomplete", l);
  }
  hipStream_t s = (hipStream_t)stream_;
  for (int g = 0; g < n; ++g) {
    StreamLane& L = lanes[g];
    L.st = st;
    L.s = g == 0 ? s : (hipStream_t)side->streams[g - 1];
    L.mel = mel + (size_t)L.b0 * T * p0.F;
    L.out = out + (size_t)L.b0 * p0.T3 * p0.d;
    L.bind((char*)workspace + L.ws_off);
    if (g > 0 && (!side->streams[g - 1] || !side->events[g])) return fail("stream step: lane %d has no stream / event", g);
  }
  // fork: the side streams start behind everything already queued on `stream`
  if (n > 1) {
    if (!side->events[0]) return fail("stream step: no fork event");
    if (hipEventRecord((hipEvent_t)side->events[0], s) != hipSuccess) return fail("stream step: event record");
    for (int g = 1; g < n; ++g)
      if (hipStreamWaitEvent(lanes[g].s, (hipEvent_t)side->events[0], 0) != hipSuccess) return fail("stream step: stream wait");
  }
  // lanes are independent (disjoint streams, workspace slices and ring slices); their kernels are queued
  // layer by layer in turn so that one lane's cache-bound attention overlaps another's weight-bound GEMMs
  int rc = 0;
  for (int g = 0; g < n && rc == 0; ++g) rc = lanes[g].front();
  for (int l = 0; l < p0.L && rc == 0; ++l)
    for (int g = 0; g < n && rc == 0; ++g) rc = lanes[g].layer(l);
  // join, also after a failed launch: `stream` must not run ahead of work already queued on the side streams
  for (int g = 1; g < n; ++g) {
    if (hipEventRecord((hipEvent_t)side->events[g], lanes[g].s) != hipSuccess ||
        hipStreamWaitEvent(s, (hipEvent_t)side->events[g], 0) != hipSuccess)
      return fail("stream step: join failed");
  }
  if (rc != 0) return -1;
  if (st->ring && launch_stream_advance(st->ring, p0.T3, st->cap, s)) return -1;
  st->frames += p0.T3;
  return 0;
}

int fcenc_greedy_decode(const float* enc_proj, int ldj, const float* pred_proj, const float* out_w, const float* out_b,
                        int B, int T, int dj, int V, int blank, int max_symbols, float clamp, int32_t* tokens, int cap,
                        int32_t* counts, int32_t* last_tokens, void* stream) {
  GreedyDecodeArgs a;
  a.enc_proj = enc_proj; a.ldj = ldj; a.pred_proj = pred_proj; a.out_w = out_w; a.out_b = out_b; a.B = B; a.T = T;
  a.dj = dj; a.V = V; a.blank = blank; a.max_symbols = max_symbols; a.clamp = clamp; a.tokens = tokens; a.cap = cap;
  a.counts = counts; a.last = last_tokens;
  return launch_greedy_decode(a, (hipStream_t)stream);
}

int fcenc_mel_spectrogram(const fcenc_mel_plan* plan, const float* wave, long long ldw, const int64_t* lengths, int B,
                          int S, float* mel, int Tmax, int64_t* frames, void* stream) {
  if (!plan) return fail("mel: null plan");
  MelArgs a;
  a.wave = wave; a.ldw = ldw; a.lengths = lengths; a.dft = plan->dft; a.win = plan->win; a.fb_start = plan->fb_start;
  a.fb_count = plan->fb_count; a.fb_w = plan->fb_w; a.mel = mel; a.frames = frames; a.B = B; a.S = S; a.Tmax = Tmax;
  a.n_fft = plan->n_fft; a.hop = plan->hop; a.kp = plan->kp; a.wpitch = plan->wpitch; a.fb_maxw = plan->fb_maxw;
  a.n_mels = plan->n_mels;
  return launch_mel_spectrogram(a, (hipStream_t)stream);
}

void* fcenc_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreate(&ev) != hipSuccess) { fail("event: create failed"); return nullptr; }
  return ev;
}
void fcenc_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int fcenc_event_record(void* ev, void* stream) {
  if (!ev || hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) return fail("event: record failed");
  return 0;
}
float fcenc_event_elapsed_ms(void* begin, void* end) {
  float ms = -1.f;
  if (!begin || !end || hipEventSynchronize((hipEvent_t)end) != hipSuccess ||
      hipEventElapsedTime(&ms, (hipEvent_t)begin, (hipEvent_t)end) != hipSuccess) {
    fail("event: elapsed failed");
    return -1.f;
  }
  return ms;
}

}  // extern "C"
