// Mel-spectrogram front-end on the matrix cores: what the reference computes per utterance with
// torchaudio.transforms.MelSpectrogram (train.py:57-61, 67-71, 122; inference.py:88-92, 294-300) and zero-pads
// into a batch (train.py:170-174), written straight in the encoder's input layout [B][T][n_mels].
//
// Per frame t of utterance b (centre = True, reflect padding, periodic Hann window of win_length centred in
// n_fft, power 2, HTK triangular filters, no normalisation):
//   x[n]   = w_b[reflect(t * hop + n - n_fft/2)],  n = 0 .. n_fft-1
//   X[k]   = sum_n win[n] x[n] e^(-2 pi i k n / n_fft),  k = 0 .. n_fft/2
//   mel[j] = sum_k fb[k][j] |X[k]|^2
// The DFT is a real matrix product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), folded once: the window is
// symmetric (win[n] = win[N - n]), cos is even and sin odd about N/2, so with m running over the first half
//   Re X[k] =  sum_m wc[m][k] (x[m] + x[N-m]),   wc[m][k] = win[m] cos(2 pi k m / N)   (halved at m = N/2)
//   Im X[k] = -sum_m ws[m][k] (x[m] - x[N-m]),   ws[m][k] = win[m] sin(2 pi k m / N)
// which halves the contraction length (201 of 400 / 512 samples carry window weight in both configurations
// the reference uses).  The matrix [204][cos columns | sin columns] comes from the host (fcenc.h), streamed
// through LDS in 12-row chunks; MFMA time hides the stream (fp32 MFMA is 1/16 of the bf16 rate).
//
// One workgroup = 64 frames of one utterance: the samples they cover are loaded once into LDS (rows of `hop`
// samples at a pitch that spreads the 16 frames of a fragment over the banks), waves 0-3 own 16 frames each and
// the cos columns (13 or 17 accumulator blocks), waves 4-7 the sin columns of the same frames (two waves per
// SIMD, so one's operand reads run under the other's MFMAs), the squared magnitudes go through LDS to the
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
constexpr int MEL_KP = 204;         // contraction length (window support / 2 + 1, padded to a multiple of 12)
constexpr int MEL_CH = 12;          // matrix rows per LDS chunk
constexpr int MEL_NCHUNK = MEL_KP / MEL_CH;

struct MelDev {
  const float* wave;
  long long ldw;
  const int64_t* lengths;
  const float* dft;
  const int32_t* fb_start;
  const int32_t* fb_count;
  const float* fb_w;
  float* mel;
  int64_t* frames;
  int S, n_fft, hop, m_lo, wpitch, fb_maxw, n_mels, Tmax, span_pitch, span_rows, n_bins, pw_pitch, tab_off;
};

template <int BC, int BS>
__global__ __launch_bounds__(MEL_THREADS, 4) void mel_kernel(const MelDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* span = reinterpret_cast<float*>(smem);                      // [span_rows][span_pitch] samples
  float* wbuf = span + p.span_rows * p.span_pitch;                   // [MEL_CH][wpitch] matrix chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int b = blockIdx.y, t0 = blockIdx.x * MEL_FT;
  const int hop = p.hop, N = p.n_fft, pitch = p.span_pitch, wpitch = p.wpitch;

  long long L = p.lengths ? p.lengths[b] : p.S;
  L = L < 1 ? 1 : (L > p.S ? p.S : L);
  const int n_frames = 1 + (int)(L / hop);
  if (blockIdx.x == 0 && tid == 0 && p.frames) p.frames[b] = n_frames;

  // ---- the samples of these 64 frames, reflected at the utterance's own ends.  Every load of a thread is
  // issued before its first LDS store (a load per loop trip would pay one HBM round trip per trip). ----
  {
    const float* w = p.wave + (size_t)b * p.ldw;
    const int E = (MEL_FT - 1) * hop + N, Li = (int)L;
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
          *reinterpret_cast<f32x4*>(span + q * pitch + (i - q * hop)) = v[j];     // hop % 4 == 0: one row
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
            span[q * pitch + (i - q * hop)] = v[j];
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
  const int chunk_vec = MEL_CH * wpitch / 4;       // float4 per chunk
  constexpr int NPRE = (MEL_CH * 544 / 4 + MEL_THREADS - 1) / MEL_THREADS;     // 4 vectors per thread
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.dft);
    f32x4* dst = reinterpret_cast<f32x4*>(wbuf);
    for (int i = tid; i < chunk_vec; i += MEL_THREADS) dst[i] = src[i];
  }
  __syncthreads();

  // two waves per SIMD: waves 0-3 take the cos columns of frames 16 * (wave & 3) .., waves 4-7 the sin columns
  // of the same frames, so one wave's LDS reads and address arithmetic run under the other's MFMAs
  const int fgrp = wave & 3;
  const bool sin_wave = wave >= 4;
  constexpr int NBLK = BC > BS ? BC : BS;
  f32x4 acc[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // forward sample n = m_lo + 4 * step + g and its mirror N - n, as (row, offset) of the span
  const int frow = 16 * fgrp + r;
  int nb = N - (p.m_lo + g);
  int qf = (p.m_lo + g) / hop, rf = (p.m_lo + g) - qf * hop;
  int qb = nb / hop, rb = nb - qb * hop;
  const float sgn = sin_wave ? -1.f : 1.f;
  const int col0 = sin_wave ? 16 * BC : 0;

  for (int c = 0; c < MEL_NCHUNK; ++c) {
    f32x4 pre[NPRE];
    if (c + 1 < MEL_NCHUNK) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.dft) + (size_t)(c + 1) * chunk_vec;
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + MEL_THREADS * i;
        if (idx < chunk_vec) pre[i] = src[idx];
      }
    }
    const float* wb = wbuf + g * wpitch + r + col0;
    // operands of step kk + 1 are read while the MFMAs of step kk run
    float bq[2][NBLK], av[2];
    auto read_step = [&](int kk, int slot) {
      const float xf = span[(frow + qf) * pitch + rf];
      const float xb = nb < N ? span[(frow + qb) * pitch + rb] : 0.f;   // n = 0 has no mirror inside the frame
      av[slot] = fmaf(sgn, xb, xf);                                      // x[m] + x[N-m] (cos) or x[m] - x[N-m] (sin)
      const float* wrow = wb + 4 * kk * wpitch;
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
        if (blk < BS || !sin_wave) bq[slot][blk] = wrow[16 * blk];
      rf += 4;
      if (rf >= hop) { rf -= hop; ++qf; }
      rb -= 4;
      nb -= 4;
      if (rb < 0) { rb += hop; --qb; }
    };
    read_step(0, 0);
#pragma unroll
    for (int kk = 0; kk < MEL_CH / 4; ++kk) {
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < MEL_CH / 4) read_step(kk + 1, (kk + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk)
        if (blk < BS || !sin_wave)
          acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk & 1], bq[kk & 1][blk], acc[blk], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // one matrix buffer (two workgroups share a CU's LDS; the other one computes across these barriers)
    __syncthreads();
    if (c + 1 < MEL_NCHUNK) {
      f32x4* dst = reinterpret_cast<f32x4*>(wbuf);
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + MEL_THREADS * i;
        if (idx < chunk_vec) dst[idx] = pre[i];
      }
      __syncthreads();
    }
  }

  // ---- |X|^2 per (frame, bin) into LDS (over the span and the matrix buffers, all reads of which are done) ----
  float* pw = reinterpret_cast<float*>(smem);                         // [64][pw_pitch]
  const int PP = p.pw_pitch, NB = p.n_bins;
  if (!sin_wave) {
#pragma unroll
    for (int blk = 0; blk < BC; ++blk) {
      const int k = 16 * blk + r;                                     // cos column = bin
      if (k < NB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pw[(16 * fgrp + 4 * g + q) * PP + k] = acc[blk][q] * acc[blk][q];
      }
    }
  }
  __syncthreads();
  if (sin_wave) {
#pragma unroll
    for (int blk = 0; blk < BS; ++blk) {
      const int k = 16 * blk + r + 1;                                 // sin column c = bin c + 1
      if (k < NB - 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pw[(16 * fgrp + 4 * g + q) * PP + k] += acc[blk][q] * acc[blk][q];
      }
    }
  }
  __syncthreads();

  // ---- filter bank: one output per lane in output order (consecutive lanes = consecutive filters of a frame), so
  // the 64 x n_mels tile is stored as it is produced, coalesced, with no staging tile ----
  {
    const int total = MEL_FT * p.n_mels, maxw = p.fb_maxw;
    float* out = p.mel + ((size_t)b * p.Tmax + t0) * p.n_mels;
    for (int e = tid; e < total; e += MEL_THREADS) {
      const int f = e / p.n_mels, j = e - f * p.n_mels;
      const int t = t0 + f;
      if (t >= p.Tmax) break;
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
  if (!a.wave || !a.dft || !a.fb_start || !a.fb_count || !a.fb_w || !a.mel) return fail("mel: null operand");
  if (a.ldw < a.S) return fail("mel: row stride %lld below the sample count %d", a.ldw, a.S);
  if (a.n_fft != 400 && a.n_fft != 512) return fail("mel: n_fft %d (400 and 512 are built)", a.n_fft);
  if (a.hop < 4 || a.hop > a.n_fft) return fail("mel: hop %d out of range", a.hop);
  if (a.m_lo < 0 || a.m_lo + MEL_KP + 3 >= a.n_fft || a.n_fft / 2 - a.m_lo + 1 > MEL_KP)
    return fail("mel: window support starting at %d does not fit %d folded samples", a.m_lo, MEL_KP);
  const int bc = (a.n_fft / 2 + 1 + 15) / 16, bs = (a.n_fft / 2 - 1 + 15) / 16;
  const int ncol = 16 * (bc + bs);
  const int wpitch = (ncol % 64 == 16 || ncol % 64 == 48) ? ncol : ncol + 16;
  if (a.wpitch != wpitch) return fail("mel: matrix pitch %d, expected %d", a.wpitch, wpitch);
  if (a.n_mels <= 0 || a.n_mels > 128 || a.fb_maxw <= 0) return fail("mel: n_mels %d / filter width %d", a.n_mels, a.fb_maxw);
  if (a.Tmax > 1 + a.S / a.hop) return fail("mel: %d frames asked of %d samples at hop %d", a.Tmax, a.S, a.hop);
  if (((uintptr_t)a.dft) & 15) return fail("mel: matrix must be 16-byte aligned");
  MelDev d;
  d.wave = a.wave; d.ldw = a.ldw; d.lengths = a.lengths; d.dft = a.dft; d.fb_start = a.fb_start; d.fb_count = a.fb_count;
  d.fb_w = a.fb_w; d.mel = a.mel; d.frames = a.frames; d.S = a.S; d.n_fft = a.n_fft; d.hop = a.hop; d.m_lo = a.m_lo;
  d.wpitch = wpitch; d.fb_maxw = a.fb_maxw; d.n_mels = a.n_mels; d.Tmax = a.Tmax;
  int pitch = a.hop;
  while (pitch % 8 != 4) ++pitch;                  // 16 frame rows x 4 samples land on 64 different banks
  d.span_pitch = pitch;
  d.span_rows = ((MEL_FT - 1) * a.hop + a.n_fft) / a.hop + 1;
  d.n_bins = a.n_fft / 2 + 1;
  d.pw_pitch = d.n_bins | 1;                       // odd: lanes = frames read a bin without conflicts
  const size_t main_bytes = ((size_t)d.span_rows * pitch + MEL_CH * wpitch) * sizeof(float);
  const size_t epi_bytes = (size_t)MEL_FT * d.pw_pitch * sizeof(float);
  const size_t work = ((main_bytes > epi_bytes ? main_bytes : epi_bytes) + 15) & ~(size_t)15;
  d.tab_off = (int)work;
  const size_t lds = work + ((size_t)a.n_mels * (a.fb_maxw | 1) + 2 * a.n_mels) * sizeof(float);
  if (lds > 160 * 1024) return fail("mel: %zu bytes of LDS needed (hop %d too small or filters too wide for this tile)", lds, a.hop);
  const void* k = a.n_fft == 400 ? reinterpret_cast<const void*>(mel_kernel<13, 13>)
                                 : reinterpret_cast<const void*>(mel_kernel<17, 16>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail("mel: cannot reserve %zu bytes of LDS", lds);
  void* args[] = {(void*)&d};
  const dim3 grid((a.Tmax + MEL_FT - 1) / MEL_FT, a.B);
  if (hipLaunchKernel(k, grid, dim3(MEL_THREADS), args, lds, stream) != hipSuccess) return fail("mel: launch failed");
  return check_launch("mel_spectrogram");
}

}  // namespace fcenc
