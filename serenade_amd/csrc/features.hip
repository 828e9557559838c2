// features.hip -- the feature front-end in front of the hot path: log-mel spectrogram and
// A-weighted loudness of serenade/bin/preprocess.py:126-203 (librosa stft / filters.mel / perceptual_weighting there).
//
// The STFT itself is one strided implicit-GEMM through srn_conv_gemm (the signal viewed as rows of 16 samples, a frame
// = n_fft / 16 taps, hop / 16 rows of stride, weights = window x DFT basis, [re | im] output columns); this file holds
// the byte-moving and elementwise ends: reflection padding, magnitude -> mel -> log, and power -> dB (with the
// per-utterance top_db floor) -> A-weighting -> amplitude -> frame mean -> log.  All HBM-bound and tiny next to the
// model (8.4 MFLOP per frame for the loudness STFT, 0.5 for the mel one).
#include <hip/hip_runtime.h>

#include "common.h"
#include "serenade_hip.h"

namespace {

// out[b][i] = x[b][reflect(i - pad)] for i < n + 2 pad, 0 up to ld (numpy.pad mode="reflect": no edge repeat)
__global__ __launch_bounds__(256) void pad_signal_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        const int n, const int pad, const int ld, const int zero) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ld; i += gridDim.x * 256) {
    float v = 0.f;
    if (i < n + 2 * pad) {
      int j = i - pad;
      const bool outside = j < 0 || j >= n;
      if (j < 0) j = -j;
      if (j >= n) j = 2 * (n - 1) - j;
      j = min(max(j, 0), n - 1);
      v = (zero && outside) ? 0.f : x[(int64_t)b * n + j];
    }
    out[(int64_t)b * ld + i] = v;
  }
}

// one workgroup per frame: |X| of the nb bins into LDS, then mel bin m = threadIdx (mel_t is (nb, n_mels): lanes read
// consecutive addresses), out = log(max(eps, dot))
__global__ __launch_bounds__(128) void logmel_kernel(const float* __restrict__ spec, const float* __restrict__ mel_t,
                                                     float* __restrict__ out, const int nb, const int ld,
                                                     const int n_mels, const float eps, const int log_mode) {
  extern __shared__ float mag[];
  const int64_t fr = blockIdx.x;
  const float* row = spec + fr * ld;
  for (int f = threadIdx.x; f < nb; f += 128) {
    const float re = row[f], im = row[nb + f];
    mag[f] = sqrtf(re * re + im * im);
  }
  __syncthreads();
  for (int m = threadIdx.x; m < n_mels; m += 128) {
    float a = 0.f;
    for (int f = 0; f < nb; ++f) a = fmaf(mag[f], mel_t[f * n_mels + m], a);
    a = fmaxf(a, eps);
    out[fr * n_mels + m] = log_mode == 10 ? log10f(a) : (log_mode == 2 ? log2f(a) : logf(a));
  }
}

// zero-fill by a kernel: a hipMemsetAsync node captured into a hipGraph replays with a corrupted fill value from the
// second replay on on this ROCm stack (profiles/r3_graph_probe_*.json), so nothing in this library issues one
__global__ void zero_u32_kernel(unsigned* __restrict__ p, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// per-utterance maximum of the power spectrogram (non-negative floats order like their bit patterns, so an integer
// atomicMax gives a bit-reproducible result); gmax must be zeroed by the caller
__global__ __launch_bounds__(256) void power_max_kernel(const float* __restrict__ spec, unsigned* __restrict__ gmax,
                                                        const int frames, const int nb, const int ld) {
  const int b = blockIdx.y;
  float m = 0.f;
  for (int fr = blockIdx.x; fr < frames; fr += gridDim.x) {
    const float* row = spec + ((int64_t)b * frames + fr) * ld;
    for (int f = threadIdx.x; f < nb; f += 256) {
      const float re = row[f], im = row[nb + f];
      m = fmaxf(m, re * re + im * im);
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(gmax + b, __float_as_uint(m));
}

// one workgroup per frame: mean over bins of 10^((max(10 log10(max(amin, p)), top) + A[f]) / 20), then log(. + 1e-5)
__global__ __launch_bounds__(256) void loudness_kernel(const float* __restrict__ spec, const float* __restrict__ aw,
                                                       const unsigned* __restrict__ gmax, float* __restrict__ out,
                                                       const int frames, const int nb, const int ld, const float amin,
                                                       const float top_db, const float add_eps) {
  __shared__ float red[4];
  const int b = blockIdx.y, fr = blockIdx.x;
  const float floor_db = 10.f * log10f(fmaxf(amin, __uint_as_float(gmax[b]))) - top_db;
  const float* row = spec + ((int64_t)b * frames + fr) * ld;
  float s = 0.f;
  for (int f = threadIdx.x; f < nb; f += 256) {
    const float re = row[f], im = row[nb + f];
    const float db = fmaxf(10.f * log10f(fmaxf(amin, re * re + im * im)), floor_db) + aw[f];
    s += exp10f(0.05f * db);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[(int64_t)b * frames + fr] = logf(((red[0] + red[1]) + (red[2] + red[3])) / nb + add_eps);
}

}  // namespace

extern "C" int srn_pad_signal(const float* x, float* out, int B, int n, int pad, int ld, int mode, void* stream) {
  SRN_CHECK_ARG(x && out && B > 0 && n > 1 && pad >= 0 && ld >= n + 2 * pad && (mode == 0 || mode == 1),
                "pad_signal: bad args");
  SRN_CHECK_ARG(mode == 1 || pad < n, "pad_signal: reflect padding of %d needs more than %d samples", pad, n);
  int bx = (ld + 255) / 256;
  bx = bx > 4096 ? 4096 : bx;
  hipLaunchKernelGGL(pad_signal_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, x, out, n, pad, ld, mode);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_logmel(const float* spec, const float* mel_t, float* out, int64_t frames, int n_bins, int ld,
                          int n_mels, float eps, int log_mode, void* stream) {
  SRN_CHECK_ARG(spec && mel_t && out && frames > 0 && frames < (1ll << 31) && n_bins > 0 && ld >= 2 * n_bins &&
                    n_mels > 0 && (log_mode == 0 || log_mode == 2 || log_mode == 10) && n_bins <= 8192,
                "logmel: bad args");
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)frames), dim3(128), (size_t)n_bins * sizeof(float),
                     (hipStream_t)stream, spec, mel_t, out, n_bins, ld, n_mels, eps, log_mode);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_loudness(const float* spec, const float* a_weight_db, unsigned* gmax_ws, float* out, int B,
                            int frames, int n_bins, int ld, float amin, float top_db, float add_eps, void* stream) {
  SRN_CHECK_ARG(spec && a_weight_db && gmax_ws && out && B > 0 && frames > 0 && n_bins > 0 && ld >= 2 * n_bins,
                "loudness: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zero_u32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, gmax_ws, B);
  hipLaunchKernelGGL(power_max_kernel, dim3(frames < 512 ? frames : 512, B), dim3(256), 0, st, spec, gmax_ws, frames,
                     n_bins, ld);
  hipLaunchKernelGGL(loudness_kernel, dim3(frames, B), dim3(256), 0, st, spec, a_weight_db, gmax_ws, out, frames,
                     n_bins, ld, amin, top_db, add_eps);
  SRN_CHECK_LAUNCH();
  return 0;
}
