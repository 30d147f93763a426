// Mel-spectrogram front-end on the matrix cores: what the reference computes per utterance with
// torchaudio.transforms.MelSpectrogram (train.py:57-61, 67-71, 122; inference.py:88-92, 294-300) and zero-pads
// into a batch (train.py:170-174), written straight in the encoder's input layout [B][T][n_mels].
//
// Per frame t of utterance b (centre = True, reflect padding, periodic Hann window of win_length centred in
// n_fft, power 2, HTK triangular filters, no normalisation):
//   x[n]   = w_b[reflect(t * hop + n - n_fft/2)],  n = 0 .. n_fft-1
//   X[k]   = sum_n win[n] x[n] e^(-2 pi i k n / n_fft),  k = 0 .. n_fft/2
//   mel[j] = sum_k fb[k][j] |X[k]|^2
// The DFT is a real matrix product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), folded twice.  With the window
// symmetric (win[n] = win[N - n]), H = N/2, Q = N/4 and, for m = 0 .. Q,
//   u[m] = win[m] (x[m] + x[N-m]),  u'[m] = u[H-m] = win[H-m] (x[H-m] + x[H+m])      (cos is even about H)
//   v[m] = win[m] (x[m] - x[N-m]),  v'[m] = v[H-m] = win[H-m] (x[H-m] - x[H+m])      (sin is odd about H)
// the even and the odd bins separate, because cos(2 pi k (H-m) / N) = (-1)^k cos(2 pi k m / N) and
// sin(2 pi k (H-m) / N) = -(-1)^k sin(2 pi k m / N):
//   Re X[2j]   = sum_m c[m] (u + u')[m] cos(2 pi 2j m / N)        Re X[2j+1] = sum_m (u - u')[m] cos(2 pi (2j+1) m / N)
//   Im X[2j]   = -sum_m (v - v')[m] sin(2 pi 2j m / N)            Im X[2j+1] = -sum_m c[m] (v + v')[m] sin(2 pi (2j+1) m / N)
// (c[Q] = 1/2, the sample that is its own mirror; the samples x[N] and the second x[H] at m = 0 are left out).
// Four products of about (Q+1) x (Q+1) instead of one of N x (N+2): a quarter of the multiply-adds.  The four
// matrices side by side, [Q+1 padded][G0 | G1 | G2 | G3 columns], and the half window come from the host
// (fcenc.h); the matrix is streamed through LDS in chunks of 8 or 12 rows.
//
// One workgroup = 64 frames of one utterance: the samples they cover are loaded once into LDS (rows of `hop`
// samples at a pitch that spreads the 16 frames of a fragment over the banks), waves 0-3 own 16 frames each and
// the two cos products (14 or 17 accumulator blocks), waves 4-7 the two sin products of the same frames (two waves
// per SIMD, so one's operand reads run under the other's MFMAs), the squared magnitudes go through LDS to the
// sparse filter bank (each filter touches a short run of bins), whose outputs are produced in output order so the
// 64 x n_mels tile leaves as one contiguous block.  Under 80 KiB of LDS and 128 registers: two workgroups per CU,
// one computes while the other loads, synchronises or runs its epilogue.  Frames beyond an utterance's own count
// are written as zeros (the collate's padding).
#include "common.h"
#include "kernels.h"

namespace fcenc {

namespace {

constexpr int MEL_FT = 64;          // frames per workgroup
constexpr int MEL_THREADS = 512;     // 8 waves: frame group = wave & 3, column group (cos | sin) = wave >> 2

struct MelDev {
  const float* wave;
  long long ldw;
  const int64_t* lengths;
  const float* dft;
  const float* win;
  const int32_t* fb_start;
  const int32_t* fb_count;
  const float* fb_w;
  float* mel;
  int64_t* frames;
  int S, n_fft, hop, kp, wpitch, fb_maxw, n_mels, Tmax, span_pitch, span_rows, n_bins, pw_pitch, tab_off;
  int spt, rs, B;      // short inputs: utterances per 64-frame tile (1: a tile is 64 frames of one utterance), span rows each
};

// B0..B3: 16-column blocks of the four products, CH: matrix rows per LDS chunk, AHEAD: operands of a step are
// read during the previous step's MFMAs (costs a second set of operand registers)
template <int B0, int B1, int B2, int B3, int CH, bool AHEAD>
__global__ __launch_bounds__(MEL_THREADS, 4) void mel_kernel(const MelDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* span = reinterpret_cast<float*>(smem);                      // [span_rows][span_pitch] samples
  float* wbuf = span + p.span_rows * p.span_pitch;                   // [CH][wpitch] matrix chunk
  float* wt = wbuf + CH * p.wpitch;                                  // [N/2 + 1] first half of the window
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  // A tile is 64 consecutive frames of utterance blockIdx.y, or -- when an utterance has at most 32 frames (streaming
  // chunks) -- the frames of p.spt whole utterances, frame f of the tile being frame f % Tmax of utterance f / Tmax.
  const int spt = p.spt;
  const int b = blockIdx.y * spt, t0 = blockIdx.x * MEL_FT;
  const int hop = p.hop, N = p.n_fft, pitch = p.span_pitch, wpitch = p.wpitch;
  __shared__ int s_frames[MEL_FT];                  // frame count of the utterance in tile slot s

  // ---- the samples of these frames, reflected at each utterance's own ends.  Every load of a thread is
  // issued before its first LDS store (a load per loop trip would pay one HBM round trip per trip). ----
  for (int sl = 0; sl < spt; ++sl) {
    const int bu = b + sl;
    if (bu >= p.B) {
      if (tid == 0) s_frames[sl] = 0;
      continue;
    }
    long long L = p.lengths ? p.lengths[bu] : p.S;
    L = L < 1 ? 1 : (L > p.S ? p.S : L);
    if (tid == 0) {
      s_frames[sl] = 1 + (int)(L / hop);
      if (blockIdx.x == 0 && p.frames) p.frames[bu] = 1 + (int)(L / hop);
    }
    float* sp = span + sl * p.rs * pitch;
    const float* w = p.wave + (size_t)bu * p.ldw;
    const int nfr = spt > 1 ? p.Tmax : MEL_FT;
    const int E = (nfr - 1) * hop + N, Li = (int)L;
    const int s0 = t0 * hop - N / 2;
    constexpr int NV = 6;                           // 16-byte vectors per thread on the interior path
    const bool interior = s0 >= 0 && s0 + E <= Li && hop % 4 == 0 && N % 8 == 0 && E <= 4 * NV * MEL_THREADS &&
                          ((reinterpret_cast<uintptr_t>(w + s0) & 15) == 0);
    if (interior) {
      const f32x4* src = reinterpret_cast<const f32x4*>(w + s0);
      f32x4 v[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i4 = tid + MEL_THREADS * j;
        if (4 * i4 < E) v[j] = src[i4];
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i = 4 * (tid + MEL_THREADS * j);
        if (i < E) {
          const int q = i / hop;
          *reinterpret_cast<f32x4*>(sp + q * pitch + (i - q * hop)) = v[j];       // hop % 4 == 0: one row
        }
      }
    } else {
      for (int base = 0; base < E; base += 8 * MEL_THREADS) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = base + tid + MEL_THREADS * j;
          int sidx = s0 + i;
          sidx = sidx < 0 ? -sidx : sidx;
          sidx = sidx >= Li ? 2 * (Li - 1) - sidx : sidx;
          sidx = sidx < 0 ? 0 : (sidx >= Li ? Li - 1 : sidx);   // frames past the utterance (never stored), too-short input
          v[j] = i < E ? w[sidx] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = base + tid + MEL_THREADS * j;
          if (i < E) {
            const int q = i / hop;
            sp[q * pitch + (i - q * hop)] = v[j];
          }
        }
      }
    }
  }
  // filter tables into LDS behind everything else
  const int fpitch = p.fb_maxw | 1;                                  // odd: lanes = filters read a tap without conflicts
  float* fbw = reinterpret_cast<float*>(smem + p.tab_off);           // [n_mels][fpitch]
  int* fbs = reinterpret_cast<int*>(fbw + p.n_mels * fpitch);        // [n_mels] start, [n_mels] count
  for (int i = tid; i < p.n_mels * p.fb_maxw; i += MEL_THREADS) {
    const int j = i / p.fb_maxw;
    fbw[j * fpitch + (i - j * p.fb_maxw)] = p.fb_w[i];
  }
  for (int i = tid; i < p.n_mels; i += MEL_THREADS) { fbs[i] = p.fb_start[i]; fbs[p.n_mels + i] = p.fb_count[i]; }
  for (int i = tid; i <= N / 2; i += MEL_THREADS) wt[i] = p.win[i];
  const int chunk_vec = CH * wpitch / 4;           // float4 per chunk
  constexpr int NPRE = (CH * 544 / 4 + MEL_THREADS - 1) / MEL_THREADS;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.dft);
    f32x4* dst = reinterpret_cast<f32x4*>(wbuf);
    for (int i = tid; i < chunk_vec; i += MEL_THREADS) dst[i] = src[i];
  }
  __syncthreads();

  // two waves per SIMD: waves 0-3 take the two cos products of frames 16 * (wave & 3) .., waves 4-7 the two sin
  // products of the same frames, so one wave's LDS reads and address arithmetic run under the other's MFMAs
  const int fgrp = wave & 3;
  const bool sin_wave = wave >= 4;
  constexpr int MA = B0 > B2 ? B0 : B2, MB = B1 > B3 ? B1 : B3;       // first / second product of a wave
  f32x4 acc[MA + MB];
#pragma unroll
  for (int i = 0; i < MA + MB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // samples m, N - m, H - m, H + m of this lane's frame as (row, offset) of the span; m = 4 * step + g
  const int Hh = N / 2;
  int frow = 16 * fgrp + r;                                             // span row of this lane's frame
  if (spt > 1) {
    const int sl = frow / p.Tmax;
    frow = sl < spt ? sl * p.rs + (frow - sl * p.Tmax) : 0;            // slots past the last utterance: any valid row
  }
  int m = g;
  int q0 = g / hop, r0 = g - q0 * hop;                                  // m      (rising)
  int q1 = (N - g) / hop, r1 = (N - g) - q1 * hop;                      // N - m  (falling)
  int q2 = (Hh - g) / hop, r2 = (Hh - g) - q2 * hop;                    // H - m  (falling)
  int q3 = (Hh + g) / hop, r3 = (Hh + g) - q3 * hop;                    // H + m  (rising)
  const float sgn = sin_wave ? -1.f : 1.f;
  const int col0 = sin_wave ? 16 * (B0 + B1) : 0;
  const int nA = sin_wave ? B2 : B0;                                    // blocks of the first product
  const int nchunk = p.kp / CH;
  // short inputs (a streaming chunk has 18 frames): frame groups wholly past the last frame skip the products
  const bool live_wave = spt > 1 ? 16 * fgrp < spt * p.Tmax : t0 + 16 * fgrp < p.Tmax;

  for (int c = 0; c < nchunk; ++c) {
    // the next chunk of the matrix: fetched now and held through the MFMAs (AHEAD), or fetched behind them
    f32x4 pre[NPRE];
    auto fetch_next = [&]() {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.dft) + (size_t)(c + 1) * chunk_vec;
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + MEL_THREADS * i;
        if (idx < chunk_vec) pre[i] = src[idx];
      }
    };
    if (AHEAD && c + 1 < nchunk) fetch_next();
    const float* wb = wbuf + g * wpitch + r + col0;
    // operands of step kk + 1 are read while the MFMAs of step kk run
    float bq[AHEAD ? 2 : 1][MA + MB], a1[AHEAD ? 2 : 1], a2[AHEAD ? 2 : 1];
    auto read_step = [&](int kk, int slot) {
      const float xa = span[(frow + q0) * pitch + r0];
      const float xc = span[(frow + q2) * pitch + r2];
      // at m = 0 the mirrors x[N] (outside the frame) and x[H] (the same sample again) are left out
      const float xb = m > 0 ? span[(frow + q1) * pitch + r1] : 0.f;
      const float xd = m > 0 ? span[(frow + q3) * pitch + r3] : 0.f;
      const float pm = wt[m] * fmaf(sgn, xb, xa);                       // u[m]  (cos) or v[m]  (sin)
      const float ph = wt[Hh - m] * fmaf(sgn, xd, xc);                  // u'[m] (cos) or v'[m] (sin)
      a1[slot] = fmaf(sgn, ph, pm);                                     // u + u' for the even cos bins, v - v' for the even sin bins
      a2[slot] = fmaf(-sgn, ph, pm);                                    // u - u' for the odd cos bins,  v + v' for the odd sin bins
      const float* wrow = wb + 4 * kk * wpitch;
#pragma unroll
      for (int blk = 0; blk < MA; ++blk)
        if (B0 == B2 || blk < nA) bq[slot][blk] = wrow[16 * blk];
#pragma unroll
      for (int blk = 0; blk < MB; ++blk) bq[slot][MA + blk] = wrow[16 * (nA + blk)];
      m += 4;
      r0 += 4; if (r0 >= hop) { r0 -= hop; ++q0; }
      r3 += 4; if (r3 >= hop) { r3 -= hop; ++q3; }
      r1 -= 4; if (r1 < 0) { r1 += hop; --q1; }
      r2 -= 4; if (r2 < 0) { r2 += hop; --q2; }
    };
    if (AHEAD && live_wave) read_step(0, 0);
#pragma unroll
    for (int kk = 0; kk < CH / 4; ++kk) {
      if (!live_wave) break;
      const int slot = AHEAD ? (kk & 1) : 0;
      __builtin_amdgcn_sched_barrier(0);
      if (AHEAD) {
        if (kk + 1 < CH / 4) read_step(kk + 1, (kk + 1) & 1);
      } else {
        read_step(kk, 0);             // the other three waves of the SIMD cover the wait
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int blk = 0; blk < MA; ++blk)
        if (B0 == B2 || blk < nA)
          acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[slot], bq[slot][blk], acc[blk], 0, 0, 0);
#pragma unroll
      for (int blk = 0; blk < MB; ++blk)
        acc[MA + blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[slot], bq[slot][MA + blk], acc[MA + blk], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // one matrix buffer (two workgroups share a CU's LDS; the other one computes across these barriers)
    if (!AHEAD && c + 1 < nchunk) fetch_next();
    __syncthreads();
    if (c + 1 < nchunk) {
      f32x4* dst = reinterpret_cast<f32x4*>(wbuf);
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + MEL_THREADS * i;
        if (idx < chunk_vec) dst[idx] = pre[i];
      }
      __syncthreads();
    }
  }

  // ---- |X|^2 per (frame, bin) into LDS (over the span and the matrix buffer, all reads of which are done) ----
  float* pw = reinterpret_cast<float*>(smem);                         // [64][pw_pitch]
  const int PP = p.pw_pitch, NB = p.n_bins, Qq = N / 4;
  auto put = [&](const f32x4& a, int bin, bool add) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* dst = pw + (16 * fgrp + 4 * g + q) * PP + bin;
      *dst = add ? *dst + a[q] * a[q] : a[q] * a[q];
    }
  };
  if (!sin_wave) {
#pragma unroll
    for (int blk = 0; blk < B0; ++blk) {
      const int j = 16 * blk + r;                                     // Re X[2j]
      if (j <= Qq) put(acc[blk], 2 * j, false);
    }
#pragma unroll
    for (int blk = 0; blk < B1; ++blk) {
      const int j = 16 * blk + r;                                     // Re X[2j+1]
      if (j < Qq) put(acc[MA + blk], 2 * j + 1, false);
    }
  }
  __syncthreads();
  if (sin_wave) {
#pragma unroll
    for (int blk = 0; blk < B2; ++blk) {
      const int j = 16 * blk + r;                                     // Im X[2j+2]
      if (j < Qq - 1) put(acc[blk], 2 * j + 2, true);
    }
#pragma unroll
    for (int blk = 0; blk < B3; ++blk) {
      const int j = 16 * blk + r;                                     // Im X[2j+1]
      if (j < Qq) put(acc[MA + blk], 2 * j + 1, true);
    }
  }
  __syncthreads();
  (void)NB;

  // ---- filter bank: one output per lane in output order (consecutive lanes = consecutive filters of a frame), so
  // the 64 x n_mels tile is stored as it is produced, coalesced, with no staging tile ----
  {
    const int total = MEL_FT * p.n_mels, maxw = p.fb_maxw;
    float* out = p.mel + ((size_t)b * p.Tmax + t0) * p.n_mels;        // the tile's frames are contiguous in the output
    const int live_frames = spt > 1 ? min(spt, p.B - b) * p.Tmax : min(MEL_FT, p.Tmax - t0);
    for (int e = tid; e < total; e += MEL_THREADS) {
      const int f = e / p.n_mels, j = e - f * p.n_mels;
      if (f >= live_frames) break;
      const int sl = spt > 1 ? f / p.Tmax : 0;
      const int t = spt > 1 ? f - sl * p.Tmax : t0 + f;
      const int n_frames = s_frames[sl];
      const int ks = fbs[j], cnt = fbs[p.n_mels + j];
      const float* fw = fbw + j * fpitch;
      const float* prow = pw + f * PP;
      float s = 0.f;
      if (maxw <= 16) {
        // fixed trip count (weights past cnt are zero, bins past the last are clamped): all 16 reads in flight
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = prow[min(ks + q, NB - 1)];
#pragma unroll
        for (int q = 0; q < 16; ++q) s = fmaf(q < cnt ? fw[min(q, maxw - 1)] : 0.f, v[q], s);
      } else {
        for (int q = 0; q < cnt; ++q) s = fmaf(fw[q], prow[ks + q], s);
      }
      out[e] = t < n_frames ? s : 0.f;
    }
  }
}

}  // namespace

int launch_mel_spectrogram(const MelArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.S <= 0 || a.Tmax <= 0) return fail("mel: empty input B=%d S=%d T=%d", a.B, a.S, a.Tmax);
  if (!a.wave || !a.dft || !a.win || !a.fb_start || !a.fb_count || !a.fb_w || !a.mel) return fail("mel: null operand");
  if (a.ldw < a.S) return fail("mel: row stride %lld below the sample count %d", a.ldw, a.S);
  if (a.n_fft != 400 && a.n_fft != 512) return fail("mel: n_fft %d (400 and 512 are built)", a.n_fft);
  if (a.hop < 4 || a.hop > a.n_fft) return fail("mel: hop %d out of range", a.hop);
  const int Q = a.n_fft / 4;
  const int ch = a.n_fft == 400 ? 8 : 12;
  const int kp = (Q + 1 + ch - 1) / ch * ch;
  if (a.kp != kp) return fail("mel: %d matrix rows, expected %d", a.kp, kp);
  const int b0 = (Q + 1 + 15) / 16, b1 = (Q + 15) / 16, b2 = (Q - 1 + 15) / 16, b3 = b1;
  const int ncol = 16 * (b0 + b1 + b2 + b3);
  const int wpitch = (ncol % 64 == 16 || ncol % 64 == 48) ? ncol : ncol + 16;
  if (a.wpitch != wpitch) return fail("mel: matrix pitch %d, expected %d", a.wpitch, wpitch);
  if (a.n_mels <= 0 || a.n_mels > 128 || a.fb_maxw <= 0) return fail("mel: n_mels %d / filter width %d", a.n_mels, a.fb_maxw);
  if (a.Tmax > 1 + a.S / a.hop) return fail("mel: %d frames asked of %d samples at hop %d", a.Tmax, a.S, a.hop);
  if (((uintptr_t)a.dft) & 15) return fail("mel: matrix must be 16-byte aligned");
  MelDev d;
  d.wave = a.wave; d.ldw = a.ldw; d.lengths = a.lengths; d.dft = a.dft; d.win = a.win; d.fb_start = a.fb_start; d.fb_count = a.fb_count;
  d.fb_w = a.fb_w; d.mel = a.mel; d.frames = a.frames; d.S = a.S; d.n_fft = a.n_fft; d.hop = a.hop; d.kp = kp;
  d.wpitch = wpitch; d.fb_maxw = a.fb_maxw; d.n_mels = a.n_mels; d.Tmax = a.Tmax;
  int pitch = a.hop;
  while (pitch % 8 != 4) ++pitch;                  // 16 frame rows x 4 samples land on 64 different banks
  d.span_pitch = pitch;
  d.span_rows = ((MEL_FT - 1) * a.hop + a.n_fft) / a.hop + 1;
  d.B = a.B; d.spt = 1; d.rs = 0;
  if (a.Tmax <= MEL_FT / 2) {
    // short utterances (streaming chunks): several whole utterances share a 64-frame tile, each with its own span
    d.rs = ((a.Tmax - 1) * a.hop + a.n_fft) / a.hop + 1;
    const int by_frames = MEL_FT / a.Tmax, by_rows = d.span_rows / d.rs;
    d.spt = by_frames < by_rows ? by_frames : by_rows;
    if (d.spt < 1) d.spt = 1;
    if (d.spt == 1) d.rs = 0;
  }
  d.n_bins = a.n_fft / 2 + 1;
  d.pw_pitch = d.n_bins | 1;                       // odd: lanes = frames read a bin without conflicts
  const size_t main_bytes = ((size_t)d.span_rows * pitch + ch * wpitch + a.n_fft / 2 + 4) * sizeof(float);
  const size_t epi_bytes = (size_t)MEL_FT * d.pw_pitch * sizeof(float);
  const size_t work = ((main_bytes > epi_bytes ? main_bytes : epi_bytes) + 15) & ~(size_t)15;
  d.tab_off = (int)work;
  const size_t lds = work + ((size_t)a.n_mels * (a.fb_maxw | 1) + 2 * a.n_mels) * sizeof(float);
  if (lds > 160 * 1024) return fail("mel: %zu bytes of LDS needed (hop %d too small or filters too wide for this tile)", lds, a.hop);
  const void* k = a.n_fft == 400 ? reinterpret_cast<const void*>(mel_kernel<7, 7, 7, 7, 8, true>)
                                 : reinterpret_cast<const void*>(mel_kernel<9, 8, 8, 8, 12, false>);
  if (reserve_lds(k, (int)lds) != hipSuccess)
    return fail("mel: cannot reserve %zu bytes of LDS", lds);
  void* args[] = {(void*)&d};
  const dim3 grid(d.spt > 1 ? 1 : (a.Tmax + MEL_FT - 1) / MEL_FT, (a.B + d.spt - 1) / d.spt);
  if (hipLaunchKernel(k, grid, dim3(MEL_THREADS), args, lds, stream) != hipSuccess) return fail("mel: launch failed");
  return check_launch("mel_spectrogram");
}

}  // namespace fcenc
