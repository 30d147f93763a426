/* libfcenc_hip.so -- C ABI of the MI355X (gfx950) FastConformer encoder kernels.
 *
 * The reference (JohanesSetiawan/conformer) has no native code and no FFI: its hot path is a
 * sequence of ATen calls issued from model.py.  Every entry point below therefore replaces a
 * span of reference Python, cited as model.py:first-last, and is bound from Python with
 * ctypes (conformer_amd/_native.py) and registered with torch.library
 * (conformer_amd/ops.py).
 *
 * Rules common to all entry points
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch allocates it) unless marked "host";
 *   - launches are enqueued on `stream` (a hipStream_t passed as void*), never synchronise,
 *     never allocate or free device memory; the library keeps no device state;
 *   - return 0 on success, a negative value on error; the message of the calling thread's last
 *     error is returned by fcenc_last_error().  Nothing throws or aborts;
 *   - dtype: FCENC_F32 (fp32 storage, exact-fp32 MFMA) or FCENC_BF16 (bf16 storage, fp32
 *     accumulation).  "T" below means that storage type.  Vectors (bias, LayerNorm affine,
 *     folded depthwise weights) are always fp32, the residual stream X is always fp32.
 */
#ifndef FCENC_H
#define FCENC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCENC_F32 0
#define FCENC_BF16 1

#define FCENC_EPI_STORE 0       /* out = A W^T + bias                                  */
#define FCENC_EPI_STORE_SILU 1  /* out = silu(A W^T + bias)                            */
#define FCENC_EPI_GLU 2         /* out = a * sigmoid(gate), W rows interleaved 16/16   */
#define FCENC_EPI_RESID 3       /* X = beta X + alpha (A W^T + bias) (+ copy in T)     */

int fcenc_version(void);
const char* fcenc_last_error(void);

/* nn.Linear / 1x1 Conv1d with fused epilogue: model.py:579-582 (FFN), :348-350 (q,k,v as one
 * GEMM over row-concatenated weights), :429 (out_proj), :520-521 (pointwise1 + GLU),
 * :541 (pointwise2), :751 (subsampling Linear), :376 (pos_proj, bias == NULL); residual adds of
 * model.py:663,667,671,674 ride in FCENC_EPI_RESID.
 * A [M][lda] T, W [N][K] T, bias [N] fp32 or NULL.  N % 128 == 0, K % 64 == 0 (bf16) / % 32 (fp32).
 * out: [M][ldo] T (N columns; N/2 for GLU).  X: fp32 [M][N]; Xc: optional T copy of X. */
int fcenc_gemm(int dtype, int epilogue, const void* A, const void* W, const float* bias, void* out,
               float* X, void* Xc, int M, int N, int K, int lda, int ldo, float alpha, float beta,
               void* stream);

/* Conv2d(3x3, stride 2, pad 1) + BatchNorm2d(eval) + SiLU on channels-last activations,
 * model.py:709-719 (conv2, conv3).  in [B][Hin][Win][Cin] T; W [Cout][9*Cin] T with
 * k = (kh*3+kw)*Cin + c and the BatchNorm scale folded in; bias [Cout] fp32 folded;
 * out [B][HoutPad][Wout][Cout] T, rows >= Hout untouched.  Hout = ceil(Hin/2), Wout = ceil(Win/2). */
int fcenc_conv3x3s2(int dtype, const void* in, const void* W, const float* bias, void* out, int B,
                    int Hin, int Win, int Cin, int Cout, int HoutPad, void* stream);

/* model.py:856-857 statistics: partial[b][s] = {sum, sum of squares} of log(mel + 1e-6) in fp64
 * over `nsplit` (<= 64) slices of utterance b; mel [B][T][F] fp32, T*F % 4 == 0. */
int fcenc_input_stats(const float* mel, double* partial, int B, int T, int F, int nsplit, void* stream);

/* model.py:856-857 (apply) + :703-707: standardise with the unbiased std finished from `partial`,
 * Conv2d(1->C, 3x3, s2, p1) + folded BatchNorm + SiLU.  w9 [9][C] fp32, bias [C] fp32,
 * out [B][ceil(T/2)][ceil(F/2)][C] T. */
int fcenc_conv1(int dtype, const float* mel, const double* partial, int nsplit, const float* w9,
                const float* bias, void* out, int B, int T, int F, int C, void* stream);

/* model.py:856-857 (apply) + :703-707 + :709-713 in ONE kernel: standardise, Conv2d(1->256, 3x3, s2, p1) + BatchNorm + SiLU,
 * Conv2d(256->256, 3x3, s2, p1) + BatchNorm + SiLU, with the first convolution produced inside the second (its
 * activation never reaches memory).  bf16, 80 mel bins and 256 channels only (fcenc_conv12_supported); other shapes
 * use fcenc_conv1 + fcenc_conv3x3s2.  A1p / W2p are the two weight sets in MFMA fragment order as written by
 * fcenc_conv12_pack from w9 [9][256] fp32, b1 [256] fp32 (fcenc_conv1's operands) and W2 [256][9*256] bf16
 * (fcenc_conv3x3s2's operand); fcenc_conv12_packed_bytes(0) / (1) give their sizes.  bias2 [256] fp32;
 * out [B][ceil(ceil(T/2)/2)][20][256] bf16, i.e. what fcenc_conv3x3s2 writes with HoutPad = Hout. */
int fcenc_conv12_supported(int dtype, int n_mels, int channels);
long long fcenc_conv12_packed_bytes(int which);
int fcenc_conv12_pack(const float* w9, const float* b1, const void* W2, void* A1p, void* W2p, int n_mels, int channels,
                      void* stream);
int fcenc_lognorm_conv12(const float* mel, const double* partial, int nsplit, const void* A1p, const void* W2p,
                         const float* bias2, void* out, int B, int T, int n_mels, int channels, void* stream);

/* model.py:715-719 for the shape the third convolution has at 80 mel bins: Conv2d(256->256, 3x3, s2, p1) + BatchNorm +
 * SiLU over 20 input columns, bf16, with the input patch of a tile staged once in LDS (fcenc_conv3_patch_supported;
 * other shapes use fcenc_conv3x3s2).  Wp: the weights W [256][9*256] bf16 of fcenc_conv3x3s2 in MFMA fragment order as
 * written by fcenc_conv3_patch_pack (fcenc_conv3_patch_packed_bytes() bytes).  in [B][Hin][20][256] bf16, bias [256]
 * fp32, out [B][HoutPad][10][256] bf16 with rows ceil(Hin/2) .. HoutPad-1 untouched: what fcenc_conv3x3s2 writes. */
int fcenc_conv3_patch_supported(int dtype, int Win, int Cin, int Cout);
long long fcenc_conv3_patch_packed_bytes(void);
int fcenc_conv3_patch_pack(const void* W, void* Wp, int channels, void* stream);
int fcenc_conv3_patch(const void* in, const void* Wp, const float* bias, void* out, int B, int Hin, int Win, int channels,
                      int HoutPad, void* stream);

/* No counterpart in the reference.  fcenc_input_stats + fcenc_conv1 in one call that ALLOCATES NOTHING and takes
 * no workspace, for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  The partial sums live in a fixed 1 MiB device array inside the library (one copy per
 * GPU, held until the process ends), so B <= 4096 utterances per call; more is rejected with an error.  Same
 * arguments and bit-identical output to the two calls above with nsplit = 16.  The device `stream` belongs to
 * must be current.  All calls on one GPU share that array: calls that overlap in time on one GPU (different
 * streams) race; calls on one stream are ordered and safe.  The first call on each device looks the array's
 * address up (no allocation either); make that first call outside a graph capture. */
int fcenc_lognorm_conv1_static(int dtype, const float* mel, const float* w9, const float* bias, void* out,
                               int B, int T, int F, int C, void* stream);

/* Training path (train.py:302-336 runs model.forward + loss.backward), fp32.
 * Backward of nn.LayerNorm(d), eps 1e-5 (model.py:516, :578, :677): X, dY, dX [M][d], g [d];
 * dX = rstd * (dY*g - mean(dY*g) - xhat * mean(dY*g*xhat)).  partial [nblocks][2][d] receives per-workgroup
 * sums of dY*xhat (weight gradient) and dY (bias gradient); the caller adds the nblocks rows up. */
int fcenc_layernorm_bwd(const float* X, const float* dY, const float* g, float* dX, float* partial, int nblocks,
                        int M, int d, void* stream);

/* nn.GLU(dim=channel) + CausalConv1d depthwise k = 9 with zero history (model.py:521-525, :236-249), token-major:
 * Y [B][T][2d] (value | gate halves) -> G [B][T][d] = value * sigmoid(gate) (kept for the backward) and
 * Z [B][T][d] = bias + sum_m w9[m][c] * G[t-8+m].  w9 [9][d], bias [d]. */
int fcenc_glu_dwconv_fwd(const float* Y, const float* w9, const float* bias, float* G, float* Z, int B, int T,
                         int d, void* stream);
/* rows of `partial` the backward below writes: one per (utterance, run of 16 frames) */
int fcenc_glu_dwconv_partial_blocks(int B, int T);
/* Backward of the above: dZ [B][T][d] -> dY [B][T][2d]; partial [blocks][10][d] receives per-workgroup sums of
 * the tap gradients (rows 0..8) and of the bias gradient (row 9). */
int fcenc_glu_dwconv_bwd(const float* Y, const float* G, const float* dZ, const float* w9, float* dY,
                         float* partial, int B, int T, int d, void* stream);

/* Backward of nn.Linear / 1x1 Conv1d (the layers fcenc_gemm's comment lists; train.py:329 loss.backward) in one
 * call: dX [M][K] = dY W, dW [N][K] = dY^T X, db [N] = column sums of dY; any of the three outputs may be NULL.
 * X [M][K], dY [M][N], W [N][K] and the outputs are fp32; the products run in `dtype` (FCENC_F32 exact,
 * FCENC_BF16: operands rounded to bf16, fp32 accumulation).  N % 128 == 0, K % 128 == 0.  The transposed (and
 * converted, zero-padded) operands the GEMM kernel needs are made inside the call in `workspace`
 * (fcenc_linear_bwd_workspace_bytes bytes, 16-byte aligned, contents undefined afterwards). */
long long fcenc_linear_bwd_workspace_bytes(int dtype, int M, int N, int K);
int fcenc_linear_bwd(int dtype, const float* X, const float* dY, const float* W, float* dX, float* dW, float* db,
                     void* workspace, int M, int N, int K, void* stream);

/* nn.BatchNorm1d in training mode followed by nn.SiLU (model.py:528-537 with self.training; the statistics run
 * over batch x time of every channel, padding frames included, as in the reference), token-major Z [M][d] fp32:
 * Y = silu((Z - mean) * rstd * g + b) with the biased batch variance and `eps`; running_mean / running_var (both
 * or neither) <- (1 - momentum) * old + momentum * {mean, unbiased variance}.  stat [4][d] receives mean, rstd,
 * g * rstd and b - mean * g * rstd and is what the backward takes; partial: fcenc_bn_partial_blocks(M) * 2 * d floats
 * of scratch.  d % 4 == 0, d <= 1024.  The caller advances num_batches_tracked. */
int fcenc_bn_partial_blocks(int M);
int fcenc_bn_silu_train_fwd(const float* Z, const float* g, const float* b, float* running_mean, float* running_var,
                            float momentum, float eps, float* Y, float* stat, float* partial, int M, int d, void* stream);
/* Backward of the above: dY [M][d] (gradient of Y) -> dZ [M][d], dg [d], db [d]. */
int fcenc_bn_silu_train_bwd(const float* Z, const float* dY, const float* g, const float* b, const float* stat, float* dZ,
                            float* dg, float* db, float* partial, int M, int d, void* stream);

/* nn.LayerNorm(d), eps 1e-5: model.py:578, :516, :677.  y1 = LN(X; g1, b1); optional
 * y2 = LN(y1; g2, b2).  Xout (fp32, may alias X) <- y1; Aout (T) <- y2 if g2 else y1;
 * compact (fp32 [B][Tvalid][d]) <- y1 without the rows t >= Tvalid of each Tp-row utterance. */
int fcenc_layernorm(int dtype, const float* X, const float* g1, const float* b1, const float* g2,
                    const float* b2, float* Xout, void* Aout, float* compact, int M, int d, int Tp,
                    int Tvalid, void* stream);

/* 1 if fcenc_ffn_fused / fcenc_ffn_pack handle this shape (bf16, d = 256, ff = 1024 or 2048), else 0:
 * the encoder descriptor must then carry the feed-forward weights in the packed order below. */
int fcenc_ffn_fused_supported(int dtype, int d, int ff);

/* One-time re-layout of nn.Linear weights (model.py:568-571) for fcenc_ffn_fused: row-major bf16
 * W1 [ff][d], W2 [d][ff] -> MFMA fragment order, 16-byte vector ((((c*8 + w)*8 + ks)*2 + i)*64 + lane)
 * with lane = 16g + r holding 8 consecutive K elements of
 *   W1p: row 256c + 32w + 16i + r of W1, columns 32ks + 8g ..
 *   W2p: row        32w + 16i + r of W2, columns 256c + 32ks + 8g ..
 * Out of place; same byte sizes as the inputs. */
int fcenc_ffn_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, int ff, void* stream);

/* Whole feed-forward module fused (see fcenc_ffn_fused_supported): model.py:577-584 with the scaled
 * residual of :663/:674, X += alpha * (W2 silu(W1 LN(X; ln) + b1) + b2), then optionally the block's
 * closing LayerNorm (:677) when post_g/post_b are given.  X fp32 [M][256] in place; W1, W2 bf16 as
 * written by fcenc_ffn_pack; Xc optional bf16 copy of the new X; compact optional fp32
 * [B][Tvalid][256] copy without the frames t >= Tvalid of each Tp-frame utterance.  Returns < 0 for
 * unsupported shapes (callers then use fcenc_layernorm + two fcenc_gemm on row-major weights).
 * With Wq != NULL the same kernel also projects the new X (the attention's concatenated q/k/v
 * Linear, model.py:366-368): Q[M][nq] = X Wq^T + bq in bf16, nq = 768, Wq [768][256] bf16 as
 * written by fcenc_qkv_pack; see fcenc_ffn_qkv_fused_supported. */
int fcenc_ffn_fused(float* X, const float* ln_g, const float* ln_b, const void* W1, const float* b1,
                    const void* W2, const float* b2, const float* post_g, const float* post_b, void* Xc,
                    float* compact, const void* Wq, const float* bq, void* Q, int nq, int M, int d,
                    int ff, int Tp, int Tvalid, float alpha, void* stream);

/* 1 if fcenc_ffn_fused can carry the q/k/v projection (fused FFN supported and heads * 64 == 256):
 * the encoder descriptor must then carry qkv_w in fcenc_qkv_pack order. */
int fcenc_ffn_qkv_fused_supported(int dtype, int d, int ff, int heads);

/* One-time re-layout of a row-major bf16 [rows][256] matrix (rows % 256 == 0) into MFMA fragment
 * order by 256-row block: 16-byte vector ((((c*8 + w)*8 + ks)*2 + i)*64 + lane), lane = 16g + r,
 * holds row 256c + 32w + 16i + r, columns 32ks + 8g .. 32ks + 8g + 7.  Out of place. */
int fcenc_rows_pack(const void* W, void* Wp, int rows, int d, void* stream);
/* The same for the q | k | v projection matrix that fcenc_ffn_fused / the block kernels take as Wq / qkv_w: fragment
 * row r of fragment i holds row 256c + 32w + 8(r >> 2) + 4i + (r & 3), so that a lane's two fragments are eight
 * consecutive output features of its token and leave the kernel as one 16-byte store. */
int fcenc_qkv_pack(const void* W, void* Wp, int rows, int d, void* stream);

/* 1 if fcenc_convmod_fused / fcenc_convmod_pack handle this shape (bf16, d = 256, depthwise kernel
 * 9), else 0: the encoder descriptor must then carry pw1_w / pw2_w in the packed order below. */
int fcenc_convmod_fused_supported(int dtype, int d, int kernel_size);

/* One-time re-layout of the pointwise conv weights (model.py:461-468, :494-497) for
 * fcenc_convmod_fused: row-major bf16 W1 [2d][d] with rows already interleaved 16 value / 16 gate
 * (the layout fcenc_gemm's GLU epilogue takes) and W2 [d][d] -> streaming order; with lane = 16g + r
 * holding 8 consecutive K elements,
 *   W1p vector ((((w*8 + ks)*2 + c)*2 + i)*64 + lane) = row 256c + 32w + 16i + r, columns 32ks + 8g ..
 *   W2p vector ((( w*8 + ks)        *2 + i)*64 + lane) = row        32w + 16i + r, columns 32ks + 8g ..
 * Out of place; same byte sizes as the inputs. */
int fcenc_convmod_pack(const void* W1, const void* W2, void* W1p, void* W2p, int d, void* stream);

/* Whole convolution module fused (see fcenc_convmod_fused_supported; non-streaming): model.py:516-544
 * with the residual of :670-671, Xout = X + W2 silu(dw9(glu(W1 LN(X; ln) + b1)) + dw_b) + b2.
 * X, Xout fp32 [M][256], M = B * Tp with Tp % 64 == 0; Xout must NOT alias X (each 64-frame tile
 * re-reads the 8 frames before it).  W1, W2 bf16 as written by fcenc_convmod_pack; b1 [512] in the
 * interleaved row order (not packed); dw_w9 [9][256] / dw_b [256] with the eval BatchNorm folded.
 * Returns < 0 for unsupported shapes (callers then use fcenc_layernorm + fcenc_gemm + fcenc_dwconv
 * + fcenc_gemm on row-major weights).
 * With O != NULL the module's input is X + O Wo^T + bo, i.e. the kernel first applies the attention's
 * out_proj and residual (model.py:429, :667) to the bf16 attention output O [M][256]; Wo [256][256]
 * bf16 as written by fcenc_rows_pack; see fcenc_convmod_oproj_supported. */
int fcenc_convmod_fused(const float* X, float* Xout, const float* ln_g, const float* ln_b,
                        const void* W1, const float* b1, const float* dw_w9, const float* dw_b,
                        const void* W2, const float* b2, const void* O, const void* Wo, const float* bo,
                        int M, int d, int Tp, void* stream);

/* 1 if fcenc_convmod_fused can carry the attention out_proj (fused module supported and
 * heads * 64 == d): the encoder descriptor must then carry out_w in fcenc_rows_pack order. */
int fcenc_convmod_oproj_supported(int dtype, int d, int kernel_size, int heads);

/* CausalConv1d(groups=d, k=9) + BatchNorm1d(eval) + SiLU: model.py:236-249, :523-538 with the
 * zero left padding of the non-streaming path.  in/out [B][Tp][d] T, w9 [9][d], bias [d] folded. */
int fcenc_dwconv(int dtype, const void* in, const float* w9, const float* bias, void* out, int B,
                 int Tp, int d, void* stream);

/* RelPositionMultiHeadAttention core, model.py:371-428, without materialising scores.
 * qkv [B*Tp][3*H*64] T (q | k | v, head h in columns h*64..h*64+63, unused head dims zero);
 * pos [2*Tp+64][H*64] T: pos_proj(pos_emb) with relative index r = (Tvalid-1)+(j-i) in row r+64
 * and zeros elsewhere; bias_u/bias_v [H*64] fp32; klen [B] int32 visible keys per utterance;
 * window > 0 enables the causal sliding window of model.py:406-412; out [B*Tp][H*64] T.
 * Tp % 64 == 0, Tvalid <= Tp. */
int fcenc_attention(int dtype, const void* qkv, const void* pos, const float* bias_u,
                    const float* bias_v, const int32_t* klen, void* out, int B, int H, int Tp,
                    int Tvalid, int window, float scale, void* stream);

/* model.py:755-760 + :863-876: out_len[b] = ceil(ceil(ceil(len/2)/2)/2) (Tvalid when lengths is
 * NULL), mask[b][t] = t >= out_len[b] (uint8, may be NULL), klen[b] = clamp(out_len, 0, Tvalid)
 * when lengths is given else Tvalid. */
int fcenc_lengths(const int64_t* lengths, int64_t* out_len, uint8_t* mask, int32_t* klen, int B,
                  int Tvalid, void* stream);

/* ---- whole encoder forward (offline path of FastConformerEncoder.forward, model.py:835-894) ----
 * Matrices are row-major in the compute dtype unless a fused kernel takes them over, in which case
 * they must be in that kernel's packed order: ff*_w1 / ff*_w2 by fcenc_ffn_pack when
 * fcenc_ffn_fused_supported; qkv_w by fcenc_qkv_pack when fcenc_ffn_qkv_fused_supported; pw1_w /
 * pw2_w by fcenc_convmod_pack when fcenc_convmod_fused_supported; out_w by fcenc_rows_pack when
 * fcenc_convmod_oproj_supported. */
typedef struct fcenc_layer {
  const float *ff1_ln_g, *ff1_ln_b;
  const void* ff1_w1; const float* ff1_b1;   /* [ff][d], [ff]                 */
  const void* ff1_w2; const float* ff1_b2;   /* [d][ff], [d]                  */
  const void* qkv_w;  const float* qkv_b;    /* [3*H*64][d], [3*H*64]         */
  const void* pos;                           /* [2*Tp+64][H*64] for this call */
  const float *bias_u, *bias_v;              /* [H*64]                        */
  const void* out_w;  const float* out_b;    /* [d][H*64], [d]                */
  const float *conv_ln_g, *conv_ln_b;
  const void* pw1_w;  const float* pw1_b;    /* [2d][d] interleaved, [2d]     */
  const float *dw_w9, *dw_b;                 /* [9][d], [d] folded BatchNorm  */
  const void* pw2_w;  const float* pw2_b;    /* [d][d], [d]                   */
  const float *ff2_ln_g, *ff2_ln_b;
  const void* ff2_w1; const float* ff2_b1;
  const void* ff2_w2; const float* ff2_b2;
  const float *ln_g, *ln_b;
} fcenc_layer;

/* Everything of a FastConformerBlock after the attention core, one kernel (model.py:667-677):
 * Xout = LN(y + 0.5 ff2(y)), y = x' + conv(x'), x' = X + O out_w^T + out_b, for bf16, d = 256,
 * heads * 64 == d, ff = 1024 or 2048, depthwise kernel 9.  X, Xout fp32 [M][d] (distinct buffers), O
 * bf16 [M][d] attention output, compact optional fp32 [B][Tvalid][d] copy of Xout without the frames
 * t >= Tvalid of each Tp-frame utterance; `layer` (host) carries packed weights as described above. */
int fcenc_block_tail(const fcenc_layer* layer, const float* X, float* Xout, const void* O,
                     float* compact, int M, int d, int ff, int heads, int Tp, int Tvalid, void* stream);

typedef struct fcenc_encoder {
  int dtype;
  int d_model, n_heads, head_dim, n_layers, ff_dim, channels, n_mels;
  int window;                                /* local_context if use_local_attn else 0 */
  const float *conv1_w9, *conv1_b;           /* [9][C], [C]                   */
  const void* conv2_w; const float* conv2_b; /* [C][9C], [C]                  */
  const void* conv3_w; const float* conv3_b;
  const void* lin_w;  const float* lin_b;    /* [d][F3*C], columns f*C + c    */
  const fcenc_layer* layers;                 /* host array of n_layers        */
  void* ev_begin;                            /* optional hipEvent_t recorded before / after the */
  void* ev_end;                              /* conv2 launch (the largest single launch), or NULL */
  void* ev_block_begin;                      /* optional hipEvent_t recorded before / after the launch that follows   */
  void* ev_block_end;                        /* the attention core of block n_layers / 2 (the kernel class with the   */
                                             /* largest share of the pass), or NULL                                   */
  const void* conv12_a1;                     /* optional: both set (fcenc_conv12_pack) = conv1 and conv2 run as the one */
  const void* conv12_w2;                     /* kernel of fcenc_lognorm_conv12 where fcenc_conv12_supported; NULL = two  */
  const void* conv3_wp;                      /* optional (fcenc_conv3_patch_pack): conv3 runs as fcenc_conv3_patch where  */
                                             /* fcenc_conv3_patch_supported; NULL = fcenc_conv3x3s2                        */
} fcenc_encoder;

/* bytes of workspace fcenc_encoder_forward needs for a (B, T) batch; 0 on error */
size_t fcenc_encoder_workspace_bytes(const fcenc_encoder* enc, int B, int T);

/* mel [B][T][n_mels] fp32 strictly positive power spectrogram, lengths [B] int64 or NULL,
 * out [B][T'][d] fp32, out_len [B] int64, mask [B][T'] uint8 or NULL; T' = ceil(T/8) after
 * three halvings.  enc and enc->layers are host memory; everything they point to is device. */
int fcenc_encoder_forward(const fcenc_encoder* enc, const float* mel, const int64_t* lengths, int B,
                          int T, float* out, int64_t* out_len, uint8_t* mask, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- cached streaming step (FastConformerEncoder.forward with cache / streaming=True) ----
 * All streams of a batch advance in lock step: every call brings the same number of new frames Tq
 * per stream (Tq <= fcenc_stream_max_chunk_frames()), tokens are packed as row = stream * Tq + i.
 *
 * Attention over the cached keys, model.py:352-366 (append, trim to the last `cap` = local_context
 * frames) + :371-428 with the reference's chunk-sized position table and its truncate / left-pad
 * fix-up of the shifted scores (:394-400).  qkv [B*Tq][3*H*64] T (q | k | v of the chunk);
 * pos [2*Tq-1][H*64] T = pos_proj(pos_emb) for a Tq-frame chunk; rings kcache [B][H][cap][64] T
 * (one row per cached frame) and vcache [B][H][cap/8][64][8] T (blocks of 8 slots; inside a block
 * one group of 8 consecutive elements per value dimension = that dimension for the block's 8
 * slots), frame t of a stream in slot t % cap, updated in place with the chunk's rows; `frames` =
 * frames every stream has seen before this chunk; out [B*Tq][H*64] T.
 * Tq <= cap <= fcenc_stream_max_cache_frames(), cap % 8 == 0.  The rings must start zero-filled
 * (slots not yet written are read and weighted by an exact 0).  bf16 runs on the matrix cores: K
 * row pieces and V groups go from global memory to MFMA fragments unchanged. */
int fcenc_stream_attention(int dtype, const void* qkv, const void* pos, const float* bias_u,
                           const float* bias_v, void* kcache, void* vcache, void* out, int B, int H,
                           int Tq, int cap, long long frames, float scale, void* stream);

/* CausalConv1d(groups=d, k=9) with its left context taken from the cache (model.py:236-249) +
 * BatchNorm1d(eval) + SiLU.  g [B*Tq][d] T GLU output of the chunk; cache [B][8][d] T: the 8 frames
 * before the chunk (zeros at the start of a stream), replaced by the last 8 frames of
 * [cache | chunk]; out [B*Tq][d] T. */
int fcenc_stream_dwconv(int dtype, const void* g, void* cache, const float* w9, const float* bias,
                        void* out, int B, int Tq, int d, void* stream);

int fcenc_stream_max_chunk_frames(void);
int fcenc_stream_max_cache_frames(void);

/* Per-layer state and the row-major copies of the matrices whose descriptor entries may be in a fused
 * kernel's packed order (the step multiplies them with fcenc_gemm). */
typedef struct fcenc_stream_layer {
  void* kcache;            /* [B][H][cap][64] T                                 */
  void* vcache;            /* [B][H][cap/8][64][8] T                            */
  void* conv_cache;        /* [B][8][d] T                                       */
  const void* pos;         /* [2*Tq-1][H*64] T for this call's chunk length     */
  const void* out_w;       /* [d][H*64] T row-major                             */
  const void* pw1_w;       /* [2d][d] T row-major, rows interleaved 16 / 16     */
  const void* pw2_w;       /* [d][d] T row-major                                */
  const void* qkv_w;       /* [3*H*64][d] T row-major; may be null: the step then keeps one workgroup per
                              64 tokens in the first feed-forward kernel instead of splitting its hidden
                              dimension over CUs when a chunk has few tokens                             */
} fcenc_stream_layer;

typedef struct fcenc_stream_state {
  int batch;                       /* concurrent streams                         */
  int cap;                         /* cached frames per stream (local_context)   */
  long long frames;                /* frames seen so far; advanced by each step  */
  fcenc_stream_layer* layers;      /* host array of enc->n_layers                */
  int* ring;                       /* null, or 3 device ints {frames % cap, min(frames + T', cap),
                                      min(frames, cap)}: the step then reads the ring position from
                                      there and moves it on itself, so that a step captured in a HIP
                                      graph can be replayed (a replay does not advance `frames`)      */
} fcenc_stream_state;

/* Optional concurrency inside one step.  The streams of a batch are independent, so the step can run
 * them as 1 + n "lanes" (contiguous slices of the batch), lane 0 on `stream` and lane i on
 * streams[i-1], with each lane's kernels queued layer by layer in turn: one lane's cache-bound
 * attention then overlaps another lane's weight-bound matrix products.  The caller owns the HIP
 * streams and events (the library keeps no device state): events[0] is recorded on `stream` and
 * waited for by every side stream before the step's first kernel, events[i] is recorded on
 * streams[i-1] after its last kernel and waited for by `stream`, so that work queued on `stream`
 * after the call sees the whole result.  Results do not depend on the number of lanes. */
typedef struct fcenc_stream_lanes {
  int n;                 /* side streams, 0..7                    */
  void* streams[7];      /* hipStream_t                           */
  void* events[8];       /* hipEvent_t: [0] fork, [1..n] joins    */
} fcenc_stream_lanes;

/* fcenc_encoder_forward with the conformer blocks run as 1 + lanes->n lanes: the front-end (lengths, statistics,
 * subsampling) runs whole on `stream`, then lane g takes utterances [B g / n, B (g + 1) / n) through all blocks on
 * its own stream (utterances are independent from there on and every buffer is row-major by token), fork / join by
 * the caller's events as described above.  One lane's attention (latency-bound, few matrix-core cycles) then
 * overlaps another lane's block tail (bound by each CU's weight stream, the same time for any number of tokens).
 * Same workspace as fcenc_encoder_forward; lanes may be NULL; results do not depend on the number of lanes. */
int fcenc_encoder_forward_lanes(const fcenc_encoder* enc, const float* mel, const int64_t* lengths, int B, int T,
                                float* out, int64_t* out_len, uint8_t* mask, void* workspace,
                                size_t workspace_bytes, void* stream, const fcenc_stream_lanes* lanes);

/* n_lanes = 1 + side streams the step will be called with */
size_t fcenc_encoder_stream_workspace_bytes(const fcenc_encoder* enc, int B, int T, int n_lanes);

/* One chunk for every stream: mel [B][T][n_mels] fp32 (normalised over the chunk itself,
 * model.py:856-857; the subsampling convs see no left context, :741-751) -> out [B][T'][d] fp32,
 * T' = fcenc_out_frames(T).  enc->layers[l].pos is not used; st->layers[l].pos must hold the table
 * for T'.  lanes may be NULL (everything on `stream`).  On success st->frames has advanced by T'. */
int fcenc_encoder_stream_step(const fcenc_encoder* enc, fcenc_stream_state* st, const float* mel, int T,
                              float* out, void* workspace, size_t workspace_bytes, void* stream,
                              const fcenc_stream_lanes* lanes);

/* ---- greedy RNN-T decode on the device (train.py:180-211 greedy_decode, inference.py:161-194) ----
 * For every utterance and every encoder frame (padded frames included, as the reference): emit
 * argmax_v clamp(out_b[v] + out_w[v] . tanh(enc_proj[b][t] + pred_proj[last]), +-clamp) until it is
 * `blank` or max_symbols symbols were emitted for the frame; `last` starts as token 0 (train.py:191,
 * inference.py:312 -- which is the blank only because the reference's blank id is 0) and becomes each
 * emitted symbol (the stateless predictor of model.py:939-961 sees only that token, so
 * pred_proj[v] = predictor_proj(predictor([v])) is a table the caller computes once per model, and
 * enc_proj = encoder_proj(encoder output) is one fcenc_gemm).  Ties take the smallest index like
 * torch.argmax.  All fp32.  enc_proj [B][T][ldj] (first dj columns used), pred_proj / out_w [V][dj],
 * out_b [V]; tokens [B][cap] int32, counts [B] int32 = symbols emitted (those beyond cap are counted
 * but not stored).  last_tokens [B] int32 may be null; otherwise utterance b starts from last_tokens[b]
 * instead of 0 and the entry is replaced by the token it ended on, which is how a stream's decoding
 * carries over from one chunk of encoder frames to the next (inference.py:311-312, 379-406).
 * dj <= 1024, V <= 4096.  One workgroup per utterance, no host round trip per symbol. */
int fcenc_greedy_decode(const float* enc_proj, int ldj, const float* pred_proj, const float* out_w,
                        const float* out_b, int B, int T, int dj, int V, int blank, int max_symbols,
                        float clamp, int32_t* tokens, int cap, int32_t* counts, int32_t* last_tokens,
                        void* stream);

/* ---- mel-spectrogram front-end (torchaudio.transforms.MelSpectrogram as called at train.py:57-61, 67-71,
 * inference.py:88-92 [n_fft 400] and inference.py:294-300 [n_fft 512, win_length 400], applied per utterance at
 * train.py:122 and zero-padded into a batch at train.py:170-174) ----
 * wave [B][ldw] fp32, lengths [B] int64 samples per utterance (null: all S) -> mel [B][Tmax][n_mels] fp32, the
 * encoder's input layout, frames [B] int64 = 1 + length / hop (may be null); frames of an utterance beyond its own
 * count are zeros.  centre = true with reflect padding at each utterance's own ends, power 2, periodic Hann
 * window of win_length centred in n_fft, HTK triangular filters without normalisation.  An utterance must be
 * longer than n_fft / 2 samples (torch.stft's own condition).  The entry point checks the padded length S only:
 * a SHORTER per-utterance lengths[b] is not an error here -- its reflection index saturates at the utterance's
 * ends and the frames come out finite but meaningless, where torch.stft (and the reference's torchaudio
 * transform) would raise.  Callers that hold the lengths on the host should reject such utterances first
 * (conformer_amd.frontend.MelSpectrogram.batch does when given host lengths).
 * The plan is host-built once per configuration (conformer_amd/frontend.py:build_plan does it).  With H = n_fft / 2,
 * Q = n_fft / 4 the kernel forms, per frame and for m = 0 .. Q, the twice-folded samples
 *   u = win[m] (x[m] + x[N-m]), u' = win[H-m] (x[H-m] + x[H+m]), v = win[m] (x[m] - x[N-m]), v' = win[H-m] (x[H-m] - x[H+m])
 * and multiplies (u + u'), (u - u'), (v - v'), (v + v') with four matrices stored side by side:
 *   win    = [H + 1] fp32: the window of win_length samples centred in n_fft (zeros outside), entries 0 .. H; it must
 *            be symmetric, win[n] = win[n_fft - n];
 *   dft    = [kp][wpitch] fp32, row m (rows past m = Q are zero), kp = Q + 1 rounded up to a multiple of 8 (n_fft 400)
 *            or 12 (n_fft 512).  Column blocks of 16, b0 = ceil((Q + 1) / 16), b1 = b3 = ceil(Q / 16),
 *            b2 = ceil((Q - 1) / 16):
 *              G0, columns [0, 16 b0):              c[m] cos(2 pi (2j) m / N),     j = column            -> Re X[2j],   j <= Q
 *              G1, next 16 b1 columns:                   cos(2 pi (2j + 1) m / N), j = column in group   -> Re X[2j+1], j < Q
 *              G2, next 16 b2 columns:                   sin(2 pi (2j + 2) m / N)                        -> Im X[2j+2], j < Q - 1
 *              G3, next 16 b3 columns:              c[m] sin(2 pi (2j + 1) m / N)                        -> Im X[2j+1], j < Q
 *            with c[Q] = 1/2 and c = 1 otherwise, unused columns zero; wpitch = 16 (b0 + b1 + b2 + b3), plus 16 when
 *            that is 0 or 32 modulo 64 (464 for n_fft 400, 528 for 512); 16-byte aligned;
 *   fb_*   = the filter bank by filter: bins fb_start[j] .. fb_start[j] + fb_count[j] - 1 carry the weights
 *            fb_w[j][0 .. fb_count[j] - 1] (row pitch fb_maxw).
 * n_fft 400 or 512, n_mels <= 128, Tmax <= 1 + S / hop. */
typedef struct fcenc_mel_plan {
  int n_fft, hop, kp, wpitch, n_mels, fb_maxw;
  const float* dft;
  const float* win;
  const int32_t* fb_start;
  const int32_t* fb_count;
  const float* fb_w;
} fcenc_mel_plan;
int fcenc_mel_spectrogram(const fcenc_mel_plan* plan, const float* wave, long long ldw, const int64_t* lengths, int B,
                          int S, float* mel, int Tmax, int64_t* frames, void* stream);

/* frames after subsampling and the padded frame count the kernels use (multiple of 64) */
int fcenc_out_frames(int T);
int fcenc_padded_frames(int T);

/* thin hipEvent helpers for callers without a HIP binding (bench.py) */
void* fcenc_event_create(void);
void fcenc_event_destroy(void* ev);
int fcenc_event_record(void* ev, void* stream);
float fcenc_event_elapsed_ms(void* begin, void* end);   /* synchronises on `end`; < 0 on error */

#ifdef __cplusplus
}
#endif
#endif
