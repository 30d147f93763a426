// Host-side launch interface of the encoder kernels (internal; the public C ABI is
// include/fcenc.h).  Every launcher validates its shapes, enqueues on `stream`, never
// synchronises or allocates, and returns 0 or a negative error code after recording a
// message retrievable through fcenc_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcenc {

int fail(const char* fmt, ...);           // records the message, returns -1
int check_launch(const char* what);       // hipGetLastError() -> 0 / fail(...)
// Raises a kernel's dynamic-LDS limit to `bytes`.  The attribute is sticky per (kernel, device), so the runtime is
// asked only the first time (and when a larger size is wanted); later launches of that kernel go straight through.
hipError_t reserve_lds(const void* kernel, int bytes);

enum Epilogue : int {
  EPI_STORE = 0,        // out = acc + bias
  EPI_STORE_SILU = 1,   // out = silu(acc + bias)
  EPI_GLU = 2,          // out[:, c] = a_c * sigmoid(gate_c); weight rows interleaved in 16-row groups
  EPI_RESID = 3,        // X = beta*X + alpha*(acc + bias); optional copy of X in the compute dtype
};

struct GemmArgs {
  const void* A = nullptr;     // activations, row-major [M][lda] (compute dtype)
  const void* W = nullptr;     // weights, row-major [N][K] (compute dtype)
  const float* bias = nullptr; // [N] fp32, may be null
  void* out = nullptr;         // EPI_STORE/SILU: [M][ldo]; EPI_GLU: [M][ldo] with N/2 columns
  float* X = nullptr;          // EPI_RESID: fp32 [M][N]
  void* Xc = nullptr;          // EPI_RESID: optional compute-dtype copy [M][N]
  int M = 0, N = 0, K = 0, lda = 0, ldo = 0;
  float alpha = 1.f, beta = 0.f;
  // a contraction cut into `splits` slices of K elements each (EPI_RESID, beta 0, no bias): slice s reads columns
  // s*K .. s*K+K-1 of A (row stride lda) and of W (row stride ldw, 0 = K) and writes plane s of X [splits][M][N]
  int ldw = 0, splits = 1;
};
int launch_gemm(int dtype, int epi, const GemmArgs& a, hipStream_t stream);

// 3x3 stride-2 pad-1 convolution over NHWC activations as an implicit GEMM
// (M = B*Hout*Wout output pixels, N = Cout, K = 9*Cin, k = (kh*3+kw)*Cin + c),
// epilogue bias + SiLU (BatchNorm pre-folded into W and bias).
struct ConvArgs {
  const void* in = nullptr;    // [B][Hin][Win][Cin]
  const void* W = nullptr;     // [Cout][9*Cin]
  const float* bias = nullptr; // [Cout]
  void* out = nullptr;         // [B][HoutPad][Wout][Cout]; rows ho >= Hout are not written
  int B = 0, Hin = 0, Win = 0, Cin = 0, Cout = 0, Hout = 0, Wout = 0, HoutPad = 0;
};
int launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t stream);
// 256 x 256-tile variant (bf16, Cout % 256 == 0), chosen by launch_conv_igemm when it applies
bool conv256_supported(int dtype, const ConvArgs& a);
int launch_conv256(const ConvArgs& a, hipStream_t stream);

// per-utterance partial sums of log(mel + 1e-6): partial[b][s] = {sum, sum of squares} (fp64)
int launch_input_stats(const float* mel, double* partial, int B, int T, int F, int nsplit, hipStream_t stream);
// conv1 (1 -> C channels) fused with the input standardisation, folded BatchNorm and SiLU
int launch_conv1(int dtype, const float* mel, const double* partial, int nsplit, const float* w9,
                 const float* bias, void* out, int B, int T, int F, int C, hipStream_t stream);

// conv1 produced inside conv2 (conv12.hip; bf16, 80 mel bins, 256 channels): the first convolution's activation never
// reaches memory.  A1p / W2p: the two weight sets in fragment order (launch_conv12_pack; conv12_a1_bytes() and
// conv12_w2_bytes() bytes); out [B][ceil(ceil(T/2)/2)][20][256] bf16
bool conv12_supported(int dtype, int n_mels, int channels);
size_t conv12_a1_bytes();
size_t conv12_w2_bytes();
int launch_conv12_pack(const float* w9, const float* b1, const void* W2, void* A1p, void* W2p, int n_mels, int channels,
                       hipStream_t stream);
int launch_conv12(const float* mel, const double* partial, int nsplit, const void* A1p, const void* W2p, const float* bias2,
                  void* out, int B, int T, int n_mels, int channels, hipStream_t stream);

// the third subsampling convolution with its input patch staged once in LDS (conv3p.hip; bf16, 256 channels, 20 input
// columns).  Wp: the weights in fragment order (launch_conv3p_pack, conv3p_packed_bytes() bytes)
bool conv3p_supported(int dtype, const ConvArgs& a);
size_t conv3p_packed_bytes();
int launch_conv3p_pack(const void* W, void* Wp, int channels, hipStream_t stream);
int launch_conv3p(const ConvArgs& a, const void* Wp, hipStream_t stream);

// statistics + conv1 with the partial sums in a fixed device array of the library (no allocation, no workspace
// argument); at most 4096 utterances per call; calls that overlap on one GPU share that array and race
int launch_lognorm_conv1_static(int dtype, const float* mel, const float* w9, const float* bias, void* out, int B, int T,
                                int F, int C, hipStream_t stream);

struct LayerNormArgs {
  const float* X = nullptr;      // [M][d] fp32 input
  const float* g1 = nullptr;     // first LayerNorm affine
  const float* b1 = nullptr;
  const float* g2 = nullptr;     // optional second LayerNorm applied to the first's output
  const float* b2 = nullptr;
  float* Xout = nullptr;         // optional fp32 [M][d]: first LayerNorm's output (may alias X)
  void* Aout = nullptr;          // optional compute-dtype [M][d]: last LayerNorm's output
  float* compact = nullptr;      // optional fp32 [B][Tvalid][d]: first LayerNorm's output, pad rows dropped
  int M = 0, d = 0, Tp = 0, Tvalid = 0;
};
int launch_layernorm(int dtype, const LayerNormArgs& a, hipStream_t stream);

// training path (fp32): LayerNorm backward; GLU + causal depthwise conv forward / backward (see train.hip).
// `partial` buffers receive per-workgroup sums of the parameter gradients that the caller adds up.
int launch_layernorm_bwd(const float* X, const float* dY, const float* g, float* dX, float* partial, int nblocks, int M,
                         int d, hipStream_t stream);
int launch_glu_dwconv_fwd(const float* Y, const float* w9, const float* bias, float* G, float* Z, int B, int T, int d,
                          hipStream_t stream);
int glu_dwconv_partial_blocks(int B, int T);
int launch_glu_dwconv_bwd(const float* Y, const float* G, const float* dZ, const float* w9, float* dY, float* partial,
                          int B, int T, int d, hipStream_t stream);

// backward of nn.Linear in one call: dX = dY W, dW = dY^T X, db = column sums of dY (any of the three may be null);
// X [M][K], dY [M][N], W [N][K], outputs fp32; products in `dtype`; ws: linear_bwd_workspace_bytes(...) bytes
size_t linear_bwd_workspace_bytes(int dtype, int M, int N, int K);
int launch_linear_bwd(int dtype, const float* X, const float* dY, const float* W, float* dX, float* dW, float* db,
                      void* ws, int M, int N, int K, hipStream_t stream);
// BatchNorm1d with batch statistics + SiLU, token-major [M][d]; stat [4][d] = mean, rstd, scale, shift (kept for
// the backward); part: bn_partial_blocks(M) x 2 x d floats of scratch
int bn_partial_blocks(int M);
int launch_bn_silu_train_fwd(const float* Z, const float* g, const float* b, float* run_mean, float* run_var,
                             float momentum, float eps, float* Y, float* stat, float* part, int M, int d,
                             hipStream_t stream);
int launch_bn_silu_train_bwd(const float* Z, const float* dY, const float* g, const float* b, const float* stat, float* dZ,
                             float* dg, float* db, float* part, int M, int d, hipStream_t stream);

// fused feed-forward block (bf16, d = 256, 384 or 512): X += alpha * (W2 silu(W1 LN(X) + b1) + b2), optionally followed
// by a closing LayerNorm (post_g/post_b); Xc: optional bf16 copy; compact: optional [B][Tvalid][d] copy
struct FfnArgs {
  float* X = nullptr;
  const float *ln_g = nullptr, *ln_b = nullptr;
  const void* W1 = nullptr;
  const float* b1 = nullptr;
  const void* W2 = nullptr;
  const float* b2 = nullptr;
  const float *post_g = nullptr, *post_b = nullptr;
  void* Xc = nullptr;
  float* compact = nullptr;
  int M = 0, d = 0, ff = 0, Tp = 0, Tvalid = 0;
  float alpha = 0.5f;
  // optional projection of the updated X by the same kernel: Q[M][nq] = X Wq^T + bq (bf16, nq = 768)
  const void* Wq = nullptr;    // [nq][d], launch_qkv_pack'ed
  const float* bq = nullptr;
  void* Q = nullptr;
  int nq = 0;
  // optional split schedule for few tokens: fp32 scratch [ff / 256][M][d]; a workgroup per (token tile, hidden
  // chunk) leaves its partial product there and a second launch adds them up (no projection in this schedule)
  float* partial = nullptr;
};
// whether the split schedule fills the chip better than one workgroup per 64 tokens
bool ffn_split_pays(int M, int d, int ff);
// row-major bf16 [rows][d] (rows % 256 == 0, d = 256) -> fragment order by 256-row block
int launch_rows_pack(const void* W, void* Wp, int rows, int d, hipStream_t stream);
// the same for the q | k | v projection matrix, fragment rows paired so that a lane owns 8 consecutive output features
int launch_qkv_pack(const void* W, void* Wp, int rows, int d, hipStream_t stream);
bool ffn_fused_supported(int dtype, int d, int ff);
int launch_ffn_fused(const FfnArgs& a, hipStream_t stream);
// d = 384 / 512 variant (ffn_wide.hip): no fused projection, no split schedule; reached through the three calls above
bool ffn_wide_supported(int dtype, int d, int ff);
int launch_ffn_wide(const FfnArgs& a, hipStream_t stream);
int launch_ffn_wide_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, hipStream_t stream);
// the fused kernel can also carry the attention's Q | K | V projection when heads * 64 == 256
bool ffn_qkv_fused_supported(int dtype, int d, int ff, int heads);
// row-major W1 [ff][d] / W2 [d][ff] (bf16) -> the fragment order launch_ffn_fused streams
int launch_ffn_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, hipStream_t stream);

// fused convolution module: Xout = X + pw2(silu(dw9(glu(pw1(LN(X))))))  (bf16, d = 256, 384 or 512, kernel 9).
// Xout must be a different buffer from X: a workgroup re-reads the 8 frames before its own.
struct ConvModArgs {
  const float* X = nullptr;
  float* Xout = nullptr;
  const float *ln_g = nullptr, *ln_b = nullptr;
  const void* W1 = nullptr;      // [2d][d], rows interleaved 16 value / 16 gate, then launch_convmod_pack'ed
  const float* b1 = nullptr;     // [2d], same interleave (not packed)
  const float* dw_w9 = nullptr;  // [9][d], BatchNorm folded
  const float* dw_b = nullptr;   // [d]
  const void* W2 = nullptr;      // [d][d], packed
  const float* b2 = nullptr;     // [d]
  // optional: the module's input is X + O Wo^T + bo (attention out_proj + residual) done by the same kernel
  const void* O = nullptr;       // [M][d] bf16 attention output
  const void* Wo = nullptr;      // [d][d], launch_rows_pack'ed
  const float* bo = nullptr;     // [d]
  int M = 0, d = 0, Tp = 0;
  // optional (with O only): the feed-forward block that follows the module, run by the same kernel on
  // the rows it has just produced; its X is ignored (Xout is used), no Xc / Wq
  const FfnArgs* ffn = nullptr;
  // optional (with ffn only): the NEXT block's first feed-forward module with its q/k/v projection
  // (Wq, bq, Q, nq set), run by the same kernel on the rows `ffn` has just produced
  const FfnArgs* next = nullptr;
};
bool convmod_fused_supported(int dtype, int d, int kernel_size);
bool convmod_oproj_supported(int dtype, int d, int kernel_size, int heads);   // O / Wo / bo (and ffn / next) usable
int launch_convmod_fused(const ConvModArgs& a, hipStream_t stream);
// d = 384 / 512 variant (convmod_wide.hip): the module alone; reached through the calls above and below
bool convmod_wide_supported(int dtype, int d, int kernel_size);
int launch_convmod_wide(const ConvModArgs& a, hipStream_t stream);
int launch_convmod_wide_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, hipStream_t stream);
// row-major W1 [2d][d] (GLU-interleaved rows) / W2 [d][d] (bf16) -> the order launch_convmod_fused streams
int launch_convmod_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, hipStream_t stream);

// streaming step: out_proj + residual + the whole conv module for packed chunk tokens in one launch (bf16, d = 256,
// heads * 64 = 256, Tq <= 16); X [B * Tq][d] is updated in place, cache [B][8][d] is read and replaced
struct StreamConvModArgs {
  float* X = nullptr;
  const void* O = nullptr;       // [B * Tq][d] attention output
  const void* Wo = nullptr;      // launch_rows_pack'ed
  const float* bo = nullptr;
  const float *ln_g = nullptr, *ln_b = nullptr;
  const void* W1 = nullptr;      // GLU-interleaved rows, launch_convmod_pack'ed
  const float* b1 = nullptr;
  const float* dw_w9 = nullptr;
  const float* dw_b = nullptr;
  const void* W2 = nullptr;      // packed
  const float* b2 = nullptr;
  void* cache = nullptr;
  int B = 0, Tq = 0, d = 0, heads = 0;
};
bool stream_convmod_supported(int dtype, int d, int heads, int Tq);
int launch_stream_convmod(const StreamConvModArgs& a, hipStream_t stream);

// causal depthwise conv (kernel 9) along time + folded BatchNorm + SiLU on [B][Tp][d]
int launch_dwconv(int dtype, const void* in, const float* w9, const float* bias, void* out,
                  int B, int Tp, int d, hipStream_t stream);

struct AttnArgs {
  const void* qkv = nullptr;   // [B*Tp][3*HD], HD = H*64: q | k | v, head h at columns h*64..
  const void* pos = nullptr;   // [2*Tp+64][HD]: projected relative-position table, row = rel index + 64
  const float* bias_u = nullptr;  // [HD]
  const float* bias_v = nullptr;  // [HD]
  const int32_t* klen = nullptr;  // [B] number of visible keys per utterance (<= Tvalid)
  void* out = nullptr;         // [B*Tp][HD]
  int B = 0, H = 0, Tp = 0, Tvalid = 0;
  int window = 0;              // > 0: causal sliding window of that many keys
  float scale = 1.f;
};
int launch_attention(int dtype, const AttnArgs& a, hipStream_t stream);

// out_len[b] = ceil(ceil(ceil(len/2)/2)/2) (or Tvalid when lengths == null), mask[b][t] = t >= out_len[b],
// klen[b] = masked ? out_len[b] : Tvalid
// optionally also clears zero_bytes bytes at zero_base + b * zero_pitch for every utterance b
int launch_lengths(const int64_t* lengths, int64_t* out_len, uint8_t* mask, int32_t* klen,
                   int B, int Tvalid, hipStream_t stream, void* zero_base = nullptr, size_t zero_pitch = 0,
                   size_t zero_bytes = 0);

// cached streaming step: tokens packed as row = stream * Tq + i (Tq frames of this chunk per stream)
struct StreamAttnArgs {
  const void* qkv = nullptr;      // [B*Tq][3*HD]: q | k | v of the chunk
  const void* pos = nullptr;      // [2*Tq-1][HD]: pos_proj(pos_emb) for a Tq-frame chunk
  const float* bias_u = nullptr;  // [HD]
  const float* bias_v = nullptr;  // [HD]
  void* kcache = nullptr;         // [B][H][cap][64] ring, frame t in slot t % cap; the chunk's rows are appended
  void* vcache = nullptr;         // [B][H][cap/8][64][8]: blocks of 8 slots, one 8-slot group per dimension
  void* out = nullptr;            // [B*Tq][HD]
  int B = 0, H = 0, Tq = 0, cap = 0;
  long long frames = 0;           // frames every stream has seen before this chunk
  const int* dyn = nullptr;       // optional device {first slot, visible keys}: overrides what `frames` implies
  float scale = 1.f;
};
int launch_stream_attention(int dtype, const StreamAttnArgs& a, hipStream_t stream);
// dyn = {first slot, visible keys, frames cached} moved on by one chunk of Tq frames
int launch_stream_advance(int* dyn, int Tq, int cap, hipStream_t stream);
// causal depthwise conv + folded BatchNorm + SiLU over [cache | chunk]; cache [B][8][d] is replaced by the
// last 8 frames of that sequence
int launch_stream_dwconv(int dtype, const void* g, void* cache, const float* w9, const float* bias, void* out, int B,
                         int Tq, int d, hipStream_t stream);
// greedy RNN-T decode of B utterances (fp32): see decode.hip
struct GreedyDecodeArgs {
  const float* enc_proj = nullptr;   // [B][T][ldj]: encoder_proj of every frame (first dj columns used)
  const float* pred_proj = nullptr;  // [V][dj]: predictor_proj(predictor(token)) per token
  const float* out_w = nullptr;      // [V][dj]
  const float* out_b = nullptr;      // [V]
  int32_t* tokens = nullptr;         // [B][cap]
  int32_t* counts = nullptr;         // [B] symbols emitted (may exceed cap: the surplus is not stored)
  int32_t* last = nullptr;           // optional [B]: token the predictor starts from, replaced by the one it ends on
  int B = 0, T = 0, dj = 0, ldj = 0, V = 0, blank = 0, max_symbols = 0, cap = 0;
  float clamp = 0.f;                 // > 0: logits clamped to +-clamp before the argmax
};
int launch_greedy_decode(const GreedyDecodeArgs& a, hipStream_t stream);

// mel-spectrogram front-end (see mel.hip): waveforms -> [B][Tmax][n_mels] fp32 in the encoder's input layout
struct MelArgs {
  const float* wave = nullptr;        // [B][ldw] fp32 samples
  long long ldw = 0;
  const int64_t* lengths = nullptr;   // [B] samples per utterance, or null (all S)
  const float* dft = nullptr;         // twice-folded DFT matrix [kp][wpitch]
  const float* win = nullptr;         // [n_fft / 2 + 1] first half of the window, centred in n_fft
  const int32_t* fb_start = nullptr;  // [n_mels] first bin of each filter
  const int32_t* fb_count = nullptr;  // [n_mels] bins of each filter
  const float* fb_w = nullptr;        // [n_mels][fb_maxw] filter weights
  float* mel = nullptr;               // [B][Tmax][n_mels]
  int64_t* frames = nullptr;          // [B] frame counts, may be null
  int B = 0, S = 0, Tmax = 0, n_fft = 0, hop = 0, kp = 0, wpitch = 0, fb_maxw = 0, n_mels = 0;
};
int launch_mel_spectrogram(const MelArgs& a, hipStream_t stream);

int stream_max_chunk_frames();    // largest Tq the two kernels above take
int stream_max_cache_frames();    // largest cap

}  // namespace fcenc
