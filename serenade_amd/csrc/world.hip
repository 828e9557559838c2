// the analysis front-end between HiFi-GAN and SiFiGAN
// (serenade/bin/ssc_postprocessing.py:142-222), hand-written for gfx950.
//
// The reference calls pyworld (WORLD: CheapTrick, D4C, code_aperiodicity), pysptk (sp2mc) and
// sifigan.utils.features (dilated_factor, SignalGenerator) -- C / Python packages that are absent from the reference
// tree: these kernels follow the published algorithms as restated in oracle/world_oracle.py (parity unpinned), plus
// the in-tree `convert_continuos_f0` / np.interp length match (pinned by tests/golden/postproc_f0.npz).
//
// Design: WORLD is float64 and ratio-of-small-differences arithmetic (group delay minus its smoothed self, sorted
// band powers 60 dB apart), so everything runs in fp64 -- the MI355X issues fp64 FMAs at the fp32 rate.  ONE
// workgroup analyses ONE frame and never leaves the CU: the F0-adaptive window is applied while gathering the
// frame's samples from the (L2-resident) waveform, the FFT runs in LDS (SoA planes, twiddles in LDS, two radix-2
// stages per pass; real data through a half-length complex transform), power
// spectrum / DC correction / rectangular smoothing (a block prefix sum) / cepstral lifter / bitonic sort of the band
// powers all work on LDS arrays, and only 513 (cepstrum) or 3 (band aperiodicity) doubles per frame go back to HBM.
#include "common.h"

namespace {

// threads per frame (template parameter NT of the frame kernels): CheapTrick 256 (three short transforms, 33 KB of
// LDS), D4C 512 (74 KB: two frames per CU -- the frame kernels are chains of short barrier-separated passes, so what
// hides their latencies is the other frame on the CU; 256 -> 512 threads at one frame per CU measured 8.0 -> 5.9 ms
// per 16 392 frames)
constexpr int NT_CHEAPTRICK = 256, NT_D4C = 512 /* at N = 2048; N / 4 in general */, NT_F0 = 256;
constexpr double kPi = 3.14159265358979323846;
constexpr double kSafeGuard = 1e-12;                        // world::kMySafeGuardMinimum
constexpr double kNoiseAfterSmoothing = 1.7716279188122702e-16;  // world::kEps * sqrt(2 / pi): E|randn| * eps

__device__ __forceinline__ int mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// sum over the workgroup, result in every thread; `red` holds NT / 64 doubles.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) s += red[i];
  return s;
}

// (cos, -sin)(2 pi m / N) for m < N / 2 from the quarter-wave table tw (N / 4 pairs): the second quarter of the half
// circle is w[m + N/4] = -i w[m] = (w.im, -w.re), which keeps the table at N / 2 doubles of LDS.
template <int LOG2N>
__device__ __forceinline__ void twiddle(const double* tw, int m, double& wr, double& wi) {
  constexpr int Q = (1 << LOG2N) / 4;
  const bool upper = m >= Q;
  m -= upper ? Q : 0;
  const double t0 = tw[2 * m], t1 = tw[2 * m + 1];
  wr = upper ? t1 : t0;
  wi = upper ? -t0 : t1;
}

// In-place decimation-in-time FFT of M = 2^LOG2M complex points held as two LDS planes.  Input in bit-reversed order,
// output in natural order.  tw is the quarter-wave table of N = M << TWSHIFT points (an M-point transform walks it in
// steps of 2^TWSHIFT).  Two radix-2 stages are done per pass over LDS (a thread takes the four points base + {0, h, 2h,
// 3h}, runs both butterfly levels in registers and writes them back), a single stage first when LOG2M is odd: the
// passes -- each a barrier plus a round trip through LDS -- are what the frame kernels are made of, so halving them
// matters more than the flops.
template <int LOG2M, int NT, int TWSHIFT = 0>
__device__ void fft_lds(double* re, double* im, const double* tw) {
  constexpr int M = 1 << LOG2M;
  constexpr int LOG2N = LOG2M + TWSHIFT;
  int s = 1;
  if (LOG2M & 1) {  // stage 1 alone: butterflies of distance 1 with w = 1
    __syncthreads();
    for (int b = threadIdx.x; b < M / 2; b += NT) {
      const int i = 2 * b;
      const double ur = re[i], ui = im[i], xr = re[i + 1], xi = im[i + 1];
      re[i] = ur + xr;
      im[i] = ui + xi;
      re[i + 1] = ur - xr;
      im[i + 1] = ui - xi;
    }
    s = 2;
  }
#pragma unroll 1
  for (; s < LOG2M; s += 2) {  // stages s and s + 1
    const int h = 1 << (s - 1);
    __syncthreads();
    for (int q = threadIdx.x; q < M / 4; q += NT) {
      const int k = q & (h - 1);
      const int base = ((q >> (s - 1)) << (s + 1)) + k;
      double w1r, w1i, w2r, w2i, w3r, w3i;
      twiddle<LOG2N>(tw, (k * (M >> s)) << TWSHIFT, w1r, w1i);              // W_{2h}^k
      twiddle<LOG2N>(tw, (k * (M >> (s + 1))) << TWSHIFT, w2r, w2i);        // W_{4h}^k
      twiddle<LOG2N>(tw, ((k + h) * (M >> (s + 1))) << TWSHIFT, w3r, w3i);  // W_{4h}^{k+h}
      const double x0r = re[base], x0i = im[base], x1r = re[base + h], x1i = im[base + h];
      const double x2r = re[base + 2 * h], x2i = im[base + 2 * h], x3r = re[base + 3 * h], x3i = im[base + 3 * h];
      // stage s: (x0, x1) and (x2, x3) with W_{2h}^k
      const double t1r = w1r * x1r - w1i * x1i, t1i = w1r * x1i + w1i * x1r;
      const double t3r = w1r * x3r - w1i * x3i, t3i = w1r * x3i + w1i * x3r;
      const double a0r = x0r + t1r, a0i = x0i + t1i, a1r = x0r - t1r, a1i = x0i - t1i;
      const double a2r = x2r + t3r, a2i = x2i + t3i, a3r = x2r - t3r, a3i = x2i - t3i;
      // stage s + 1: (a0, a2) with W_{4h}^k, (a1, a3) with W_{4h}^{k+h}
      const double u2r = w2r * a2r - w2i * a2i, u2i = w2r * a2i + w2i * a2r;
      const double u3r = w3r * a3r - w3i * a3i, u3i = w3r * a3i + w3i * a3r;
      re[base] = a0r + u2r;
      im[base] = a0i + u2i;
      re[base + 2 * h] = a0r - u2r;
      im[base + 2 * h] = a0i - u2i;
      re[base + h] = a1r + u3r;
      im[base + h] = a1i + u3i;
      re[base + 3 * h] = a1r - u3r;
      im[base + 3 * h] = a1i - u3i;
    }
  }
  __syncthreads();
}

template <int LOG2N>
__device__ __forceinline__ int brev(int n) { return (int)(__brev((unsigned)n) >> (32 - LOG2N)); }

// Transform of N = 2^LOG2N REAL samples x at half the cost: z[m] = x[2m] + i x[2m+1] goes through an N/2-point complex
// transform, one more pass separates X[k] = E[k] + w_N^k O[k], k = 0 .. N/2 (E, O: transforms of the even / odd
// samples, recovered from Z[k] and conj Z[N/2 - k]).  The caller stores sample j at rfft_slot(j): plane (j & 1),
// position bit-reversed over N/2.  Five of D4C's seven transforms and all three of CheapTrick's are of real data.
template <int LOG2N>
__device__ __forceinline__ int rfft_slot(int j) { return brev<LOG2N - 1>(j >> 1); }

template <int LOG2N, int NT>
__device__ void rfft_lds(double* re, double* im, const double* tw) {
  constexpr int M = 1 << (LOG2N - 1);
  fft_lds<LOG2N - 1, NT, 1>(re, im, tw);
  for (int k = threadIdx.x; k <= M / 2; k += NT) {
    const int kk = (M - k) & (M - 1);
    const double ar = re[k], ai = im[k], br = re[kk], bi = im[kk];
    const double er = 0.5 * (ar + br), ei = 0.5 * (ai - bi);
    const double orr = 0.5 * (ai + bi), oi = -0.5 * (ar - br);
    double wr, wi;
    twiddle<LOG2N>(tw, k, wr, wi);
    const double pr = wr * orr - wi * oi, pi = wr * oi + wi * orr;
    re[k] = er + pr;
    im[k] = ei + pi;
    re[M - k] = er - pr;      // X[N/2 - k] = conj(E[k] - w^k O[k])
    im[M - k] = -(ei - pi);
  }
  __syncthreads();
}

enum { WIN_CHEAPTRICK = 0, WIN_HANNING = 1, WIN_BLACKMAN = 2 };

// F0-adaptive windowing (cheaptrick.cpp / d4c.cpp GetWindowedWaveform): gathers 2*hwl+1 samples around `position`
// (edge samples repeated), applies the window, removes the window-weighted mean, optionally scales to unit energy,
// and leaves the frame ready for its transform: with `ramp` (D4C's centroid) re = frame, im = (n + 1) * frame in
// bit-reversed order -- the frame and its time-weighted copy ride in ONE complex FFT (fft_lds); without, the real
// frame packed for rfft_lds.  Scratch: scr[0..N) holds the window,
// the im plane the products on the way (natural order); the final values pass through registers on their way to
// the bit-reversed positions.  Returns the number of windowed samples, or 0 when unit energy was requested on an
// all-zero frame.
template <int LOG2N, int NT>
__device__ int windowed_frame(const double* __restrict__ x, int x_len, int fs, double f0, double position, int kind,
                              double ratio, bool unit_energy, bool ramp, double* re, double* im, double* scr,
                              double* red) {
  constexpr int N = 1 << LOG2N;
  constexpr int NI = (N + NT - 1) / NT;
  int hwl = kind == WIN_CHEAPTRICK ? mround(1.5 * fs / f0) : mround(ratio * fs / f0 / 2.0);
  hwl = clampi(hwl, 0, (N - 1) / 2);
  const int n = 2 * hwl + 1;
  const int origin = mround(position * fs + 0.001);
  double ww = 0.0;
  __syncthreads();  // the planes may still be read by the caller's previous step
  for (int j = threadIdx.x; j < n; j += NT) {
    const int base = j - hwl;
    double w;
    if (kind == WIN_CHEAPTRICK) {
      const double pos = base / 1.5 / fs;
      w = 0.5 * cos(kPi * pos * f0) + 0.5;
    } else {
      const double pos = (2.0 * base / ratio) / fs;
      const double c1 = cos(kPi * pos * f0);
      w = kind == WIN_HANNING ? 0.5 * c1 + 0.5 : 0.42 + 0.5 * c1 + 0.08 * cos(kPi * pos * f0 * 2);
    }
    scr[j] = w;
    im[j] = x[clampi(origin + base, 0, x_len - 1)];
    ww += w * w;
  }
  double wnorm = 1.0;  // CheapTrick's window has unit energy; D4C's windows are used as they are
  if (kind == WIN_CHEAPTRICK) wnorm = sqrt(block_sum<NT>(ww, red));
  double sw = 0.0, sxw = 0.0;
  for (int j = threadIdx.x; j < n; j += NT) {  // every thread revisits its own j: no barrier needed in between
    const double w = scr[j] / wnorm;
    scr[j] = w;
    const double v = im[j] * w;
    im[j] = v;
    sw += w;
    sxw += v;
  }
  sw = block_sum<NT>(sw, red);
  sxw = block_sum<NT>(sxw, red);
  const double coef = sxw / sw;
  double vals[NI];
  double pw = 0.0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = threadIdx.x + i * NT;
    vals[i] = j < n ? im[j] - scr[j] * coef : 0.0;
    pw += vals[i] * vals[i];
  }
  double escale = 1.0;
  if (unit_energy) {
    pw = block_sum<NT>(pw, red);
    if (!(pw > 0.0)) return 0;
    escale = sqrt(pw);
  }
  __syncthreads();  // all natural-order reads of im are done
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = threadIdx.x + i * NT;
    if (j < N) {
      const double v = unit_energy ? vals[i] / escale : vals[i];
      if (ramp) {  // complex transform of (frame, time-weighted frame)
        const int r = brev<LOG2N>(j);
        re[r] = v;
        im[r] = v * (j + 1.0);
      } else {  // real transform (rfft_lds)
        ((j & 1) ? im : re)[rfft_slot<LOG2N>(j)] = v;
      }
    }
  }
  return n;  // the transforms start with a barrier
}

// interp1Q (common.cpp): y on the grid x0 + shift * k, linear, y's last difference taken as 0.
__device__ __forceinline__ double interp1q(double x0, double shift, const double* y, int y_len, double xi) {
  const double pos = (xi - x0) / shift;
  int base = (int)pos;
  const double frac = pos - base;
  base = clampi(base, 0, y_len - 1);
  const double dy = base + 1 < y_len ? y[base + 1] - y[base] : 0.0;
  return y[base] + dy * frac;
}

// DCCorrection (common.cpp): the power below F0 gets the mirror image of the power between F0 and 0 added.
template <int NT>
__device__ void dc_correction(double* a, int half, double f0, int fs, int N, double* scr) {
  int upper = 2 + (int)(f0 * N / fs);
  upper = clampi(upper, 2, half);
  const int n_rep = upper - 1;
  __syncthreads();
  for (int k = threadIdx.x; k < n_rep; k += NT)
    scr[k] = interp1q(f0, -(double)fs / N, a, upper + 1, (double)k * fs / N);
  __syncthreads();
  for (int k = threadIdx.x; k < n_rep; k += NT) a[k] += scr[k];
  __syncthreads();
}

// inclusive prefix sum of a[0..M) in place
template <int NT>
__device__ void block_cumsum(double* a, int M, double* red) {
  const int per = (M + NT - 1) / NT;
  const int lo = min((int)threadIdx.x * per, M), hi = min(lo + per, M);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) s += a[i];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) red[w] = inc;
  __syncthreads();
  double run = inc - s;
  for (int i = 0; i < w; ++i) run += red[i];
  for (int i = lo; i < hi; ++i) {
    run += a[i];
    a[i] = run;
  }
  __syncthreads();
}

// LinearSmoothing (common.cpp): mean of the (mirrored, piecewise-constant) spectrum over [f - width/2, f + width/2]
// as a difference of its running integral.  in -> out (may alias), both half+1 long; scr: half + 2*boundary + 1.
template <int NT>
__device__ void linear_smoothing(const double* in, double* out, int half, double width, int fs, int N, double add,
                                 double* scr, double* red) {
  int boundary = (int)(width * N / fs) + 1;
  boundary = clampi(boundary, 1, N / 4 + 1);  // width <= F0 <= fs / 4 (the kernels clamp F0): M <= N + 4 fits scr
  const int M = half + 2 * boundary + 1;
  const double step = (double)fs / N;
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += NT) {
    double v;
    if (m < boundary) v = in[boundary - m];
    else if (m < half + boundary) v = in[m - boundary];
    else v = in[half - (m - (half + boundary))];
    scr[m] = v * fs / N;
  }
  __syncthreads();
  block_cumsum<NT>(scr, M, red);
  const double origin = -(boundary - 0.5) * fs / N;
  for (int k = threadIdx.x; k <= half; k += NT) {
    const double ax = (double)k / N * fs - width / 2.0;
    const double low = interp1q(origin, step, scr, M, ax);
    const double high = interp1q(origin, step, scr, M, ax + width);
    out[k] = (high - low) / width + add;
  }
  __syncthreads();
}

struct Frame {
  const double* x;
  int x_len;
  double f0, t;
  int64_t row;  // b * f_bs + f
  bool valid;
};

__device__ __forceinline__ Frame load_frame(const SrnWorldParams& p) {
  Frame fr;
  const int f = blockIdx.x, b = blockIdx.y;
  fr.valid = f < p.n_frames[b];
  fr.x = p.x + (int64_t)b * p.x_bs;
  fr.x_len = p.x_len[b];
  fr.row = (int64_t)b * p.f_bs + f;
  fr.f0 = fr.valid ? p.f0[fr.row] : 0.0;
  fr.t = fr.valid ? p.t[fr.row] : 0.0;
  return fr;
}

template <int LOG2N, int NT>
__device__ __forceinline__ void load_twiddles(const double* __restrict__ g, double* tw) {
  constexpr int N = 1 << LOG2N;
  for (int i = threadIdx.x; i < N / 2; i += NT) tw[i] = g[i];  // the first quarter of the host's N/2-pair table
}

// ------------------------------------------------------------------------------------------------ CheapTrick
template <int LOG2N, int NT>
__global__ __launch_bounds__(NT) void cheaptrick_kernel(const SrnWorldParams p) {
  constexpr int N = 1 << LOG2N, HALF = N / 2;
  extern __shared__ __attribute__((aligned(16))) double lds_w[];
  double* re = lds_w;            // N
  double* im = re + N;           // N
  double* tw = im + N;           // N / 2 (quarter-wave table)
  double* scr = tw + N / 2;      // N + 8
  double* pw = scr + N + 8;      // HALF + 1
  double* red = pw + HALF + 1;   // 2 NT / 64
  const Frame fr = load_frame(p);
  if (!fr.valid) return;
  load_twiddles<LOG2N, NT>(p.twiddle, tw);
  const int fs = p.fs;
  // f0 at or below the floor (and anything that is not a usable F0) is analysed as kDefaultF0 (cheaptrick.cpp)
  double f0 = fr.f0 <= p.f0_floor ? 500.0 : fr.f0;
  if (!(f0 < 0.25 * fs)) f0 = 0.25 * fs;
  const int n = windowed_frame<LOG2N, NT>(fr.x, fr.x_len, fs, f0, fr.t, WIN_CHEAPTRICK, 0.0, false, false, re, im, scr, red);
  rfft_lds<LOG2N, NT>(re, im, tw);
  // power spectrum (+ the expected power of WORLD's 1e-12 * randn() safeguard) with DC correction
  const double floor_power = n * kSafeGuard * kSafeGuard;
  for (int k = threadIdx.x; k <= HALF; k += NT) pw[k] = re[k] * re[k] + im[k] * im[k] + floor_power;
  dc_correction<NT>(pw, HALF, f0, fs, N, scr);
  linear_smoothing<NT>(pw, pw, HALF, f0 * 2.0 / 3.0, fs, N, kNoiseAfterSmoothing, scr, red);
  // cepstrum of the symmetric log spectrum
  for (int k = threadIdx.x; k <= HALF; k += NT) pw[k] = log(pw[k]);
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += NT) ((j & 1) ? im : re)[rfft_slot<LOG2N>(j)] = pw[j <= HALF ? j : N - j];
  rfft_lds<LOG2N, NT>(re, im, tw);
  // smoothing lifter sinc(f0 q) and compensation lifter (1 - 2 q1) + 2 q1 cos(2 pi f0 q), then / N
  const double q1 = p.q1;
  double* ceps = p.out1 ? p.out1 + (int64_t)blockIdx.y * p.out1_bs + (int64_t)blockIdx.x * p.ld_out1 : nullptr;
  for (int k = threadIdx.x; k <= HALF; k += NT) {
    double sm = 1.0, comp = (1.0 - 2.0 * q1) + 2.0 * q1;
    if (k > 0) {
      const double quef = (double)k / fs;
      sm = sin(kPi * f0 * quef) / (kPi * f0 * quef);
      comp = (1.0 - 2.0 * q1) + 2.0 * q1 * cos(2.0 * kPi * quef * f0);
    }
    const double v = re[k] * sm * comp / N;
    pw[k] = v;
    if (ceps) ceps[k] = v;
  }
  if (!p.out0) return;
  // spectral envelope = exp(c2r(lifted cepstrum)): the transform of the symmetric sequence is real
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += NT) ((j & 1) ? im : re)[rfft_slot<LOG2N>(j)] = pw[j <= HALF ? j : N - j];
  rfft_lds<LOG2N, NT>(re, im, tw);
  double* sp = p.out0 + (int64_t)blockIdx.y * p.out0_bs + (int64_t)blockIdx.x * p.ld_out0;
  for (int k = threadIdx.x; k <= HALF; k += NT) sp[k] = exp(re[k]);
}

// ------------------------------------------------------------------------------------------------ D4C
// Ascending bitonic sort of the 2 NT doubles a[0 .. 2 NT) in LDS, left in REGISTERS: thread t ends up with the sorted
// elements 2t and 2t + 1.  Compare-exchange distance 1 stays inside the thread, distances 2 .. 64 are lane exchanges
// inside a wave (no barrier), only distances >= 128 (6 of the 55 steps at 1024 elements) go through LDS.
template <int NT>
__device__ __forceinline__ void bitonic_sort_pairs(double* a, double& v0, double& v1) {
  constexpr int n = 2 * NT;
  const int i0 = 2 * threadIdx.x;
  v0 = a[i0];
  v1 = a[i0 + 1];
  for (int k = 2; k <= n; k <<= 1) {
    const bool asc = (i0 & k) == 0;  // the same for both elements (k >= 2)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 1) {
        const double lo = fmin(v0, v1), hi = fmax(v0, v1);
        v0 = asc ? lo : hi;
        v1 = asc ? hi : lo;
        continue;
      }
      double p0, p1;
      if (j <= 64) {
        p0 = __shfl_xor(v0, j >> 1, 64);
        p1 = __shfl_xor(v1, j >> 1, 64);
      } else {
        __syncthreads();
        a[i0] = v0;
        a[i0 + 1] = v1;
        __syncthreads();
        p0 = a[i0 ^ j];
        p1 = a[(i0 + 1) ^ j];
      }
      const bool take_min = ((i0 & j) == 0) == asc;
      v0 = take_min ? fmin(v0, p0) : fmax(v0, p0);
      v1 = take_min ? fmin(v1, p1) : fmax(v1, p1);
    }
  }
}

template <int LOG2N, int NT>
__global__ __launch_bounds__(NT, 2) void d4c_kernel(const SrnWorldParams p) {
  constexpr int N = 1 << LOG2N, HALF = N / 2;
  extern __shared__ __attribute__((aligned(16))) double lds_w[];
  double* re = lds_w;
  double* im = re + N;
  double* tw = im + N;             // N / 2 (quarter-wave table)
  double* scr = tw + N / 2;        // N + 8
  double* pw = scr + N + 8;        // HALF + 1
  double* gd = pw + HALF + 1;      // HALF + 1: centroid, then group delay
  double* red = gd + HALF + 1;     // 2 NT / 64
  constexpr int NW = NT / 64;
  __shared__ int s_arg;
  const Frame fr = load_frame(p);
  if (!fr.valid) return;
  const int fs = p.fs, n_bands = p.n_bands;
  double* out = p.out0 + (int64_t)blockIdx.y * p.out0_bs + (int64_t)blockIdx.x * p.ld_out0;
  const double unvoiced = p.unvoiced_db;
  // every exit below is uniform over the workgroup
  if (!(fr.f0 > 0.0)) {  // f0 == 0 (and NaN): WORLD's initial aperiodicity 1 - 1e-12
    if (threadIdx.x < n_bands) out[threadIdx.x] = unvoiced;
    return;
  }
  load_twiddles<LOG2N, NT>(p.twiddle, tw);
  double f0 = fr.f0 < 0.25 * fs ? fr.f0 : 0.25 * fs;

  // ---- D4C Love Train: share of the 100 Hz .. 4 kHz power in 100 Hz .. 7.9 kHz, Blackman window of 3 periods
  {
    const double cur = fmax(f0, 40.0);
    windowed_frame<LOG2N, NT>(fr.x, fr.x_len, fs, cur, fr.t, WIN_BLACKMAN, 3.0, false, false, re, im, scr, red);
    rfft_lds<LOG2N, NT>(re, im, tw);
    const int b0 = (int)ceil(100.0 * N / fs), b1 = (int)ceil(4000.0 * N / fs), b2 = (int)ceil(7900.0 * N / fs);
    double s1 = 0.0, s2 = 0.0;
    for (int k = b0 + 1 + threadIdx.x; k <= min(b2, HALF); k += NT) {
      const double v = re[k] * re[k] + im[k] * im[k];
      s2 += v;
      if (k <= b1) s1 += v;
    }
    s1 = block_sum<NT>(s1, red);
    s2 = block_sum<NT>(s2, red);
    if (!(s2 > 0.0) || s1 / s2 <= p.threshold) {
      if (threadIdx.x < n_bands) out[threadIdx.x] = unvoiced;
      return;
    }
  }
  f0 = fmax(f0, 47.0);  // world::kFloorF0D4C

  // ---- static centroid: two frames a quarter period before / after, energy-normalised, Blackman of 4 periods
  for (int k = threadIdx.x; k <= HALF; k += NT) gd[k] = 0.0;
  for (int side = 0; side < 2; ++side) {
    const double pos = fr.t + (side == 0 ? -0.25 : 0.25) / f0;
    const int n = windowed_frame<LOG2N, NT>(fr.x, fr.x_len, fs, f0, pos, WIN_BLACKMAN, 4.0, true, true, re, im, scr, red);
    if (n == 0) {  // digital silence under a voiced F0: no periodicity to measure
      if (threadIdx.x < n_bands) out[threadIdx.x] = unvoiced;
      return;
    }
    fft_lds<LOG2N, NT>(re, im, tw);
    // Z = S1 + i S2 (S1: frame, S2: time-weighted frame): S1 = (Z[k] + conj Z[N-k]) / 2, S2 = (Z[k] - conj Z[N-k]) / 2i
    for (int k = threadIdx.x; k <= HALF; k += NT) {
      const int m = (N - k) & (N - 1);
      const double zr = re[k], zi = im[k], cr = re[m], ci = -im[m];
      const double s1r = 0.5 * (zr + cr), s1i = 0.5 * (zi + ci);
      const double dr = zr - cr, di = zi - ci;
      const double s2r = 0.5 * di, s2i = -0.5 * dr;
      gd[k] += s2r * s1r + s1i * s2i;
    }
    __syncthreads();
  }
  dc_correction<NT>(gd, HALF, f0, fs, N, scr);

  // ---- smoothed power spectrum: Hanning window of 4 periods, DC correction, smoothing over one F0
  windowed_frame<LOG2N, NT>(fr.x, fr.x_len, fs, f0, fr.t, WIN_HANNING, 4.0, false, false, re, im, scr, red);
  rfft_lds<LOG2N, NT>(re, im, tw);
  for (int k = threadIdx.x; k <= HALF; k += NT) pw[k] = re[k] * re[k] + im[k] * im[k];
  dc_correction<NT>(pw, HALF, f0, fs, N, scr);
  linear_smoothing<NT>(pw, pw, HALF, f0, fs, N, 0.0, scr, red);

  // ---- static group delay minus its own smoothed version
  for (int k = threadIdx.x; k <= HALF; k += NT) gd[k] = gd[k] / pw[k];
  linear_smoothing<NT>(gd, gd, HALF, f0 / 2.0, fs, N, 0.0, scr, red);
  linear_smoothing<NT>(gd, pw, HALF, f0, fs, N, 0.0, scr, red);  // the power spectrum is spent: its plane takes this
  for (int k = threadIdx.x; k <= HALF; k += NT) gd[k] -= pw[k];
  __syncthreads();

  // ---- per band: Nuttall-windowed group delay -> power spectrum -> share of everything but the `boundary`+1
  //      largest bins (sorted ascending, summed from the small end as WORLD does)
  const int wl = p.band_window_len, hw = wl / 2;
  const int boundary = mround(N * 8.0 / wl);
  const int n_small = HALF - boundary;  // c[HALF - boundary - 1] of the ascending cumulative sum
  for (int band = 0; band < n_bands; ++band) {
    const int center = (int)(3000.0 * (band + 1) * N / fs);
    for (int j = threadIdx.x; j < N; j += NT) {
      const int src = center - hw + j;
      ((j & 1) ? im : re)[rfft_slot<LOG2N>(j)] = (j < wl && src >= 0 && src <= HALF) ? gd[src] * p.band_window[j] : 0.0;
    }
    rfft_lds<LOG2N, NT>(re, im, tw);
    // powers into scr[0..HALF]; the largest one is taken out so that HALF (a power of two) values get sorted
    double best = -1.0;
    int arg = 0;
    for (int k = threadIdx.x; k <= HALF; k += NT) {
      const double v = re[k] * re[k] + im[k] * im[k];
      scr[k] = v;
      if (v > best) {
        best = v;
        arg = k;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ob > best || (ob == best && oa < arg)) {
        best = ob;
        arg = oa;
      }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6] = best;
      red[NW + (threadIdx.x >> 6)] = (double)arg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int w = 0;
      for (int i = 1; i < NW; ++i)
        if (red[i] > red[w]) w = i;
      s_arg = (int)red[NW + w];
    }
    __syncthreads();
    const int amax = s_arg;
    const double vmax = scr[amax];
    __syncthreads();
    if (threadIdx.x == 0) scr[amax] = scr[HALF];  // drop the maximum: HALF values remain in scr[0..HALF)
    static_assert(HALF == 2 * NT, "bitonic_sort_pairs holds two elements per thread");
    double e0, e1;
    __syncthreads();
    bitonic_sort_pairs<NT>(scr, e0, e1);
    double small = 0.0, rest = 0.0;
    {
      const int i0 = 2 * threadIdx.x;
      if (i0 < n_small) small += e0; else rest += e0;
      if (i0 + 1 < n_small) small += e1; else rest += e1;
    }
    small = block_sum<NT>(small, red);
    rest = block_sum<NT>(rest, red);
    const double coarse = 10.0 * log10(small / (small + rest + vmax)) + (f0 - 100.0) / 50.0;
    if (threadIdx.x == 0) out[band] = fmin(0.0, coarse);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ projections
// out[r, k] = sum_q mat_t[q, k] * g(in[r, q]), g = log or identity: sp2mc as ONE matrix (irfft, c0 / 2 and SPTK's
// freqt recursion are linear), and the same for the liftered cepstrum CheapTrick leaves behind.
constexpr int PR = 8;  // rows per workgroup
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ in, int64_t rows, int K, int ld_in,
                                                     const double* __restrict__ mat_t, int n_out, int take_log,
                                                     double* __restrict__ out, int ld_out) {
  extern __shared__ __attribute__((aligned(16))) double lds_p[];  // PR x K
  const int64_t r0 = (int64_t)blockIdx.x * PR;
  for (int i = threadIdx.x; i < PR * K; i += 256) {
    const int r = i / K, q = i - r * K;
    double v = 0.0;
    if (r0 + r < rows) {
      v = in[(r0 + r) * ld_in + q];
      if (take_log) v = log(v);
    }
    lds_p[i] = v;
  }
  __syncthreads();
  const int k = threadIdx.x & 63, g = threadIdx.x >> 6;  // 4 groups x 2 rows
  if (k >= n_out) return;
  double a0 = 0.0, a1 = 0.0;
  const double* x0 = lds_p + (2 * g) * K;
  const double* x1 = x0 + K;
  for (int q = 0; q < K; ++q) {
    const double m = mat_t[(int64_t)q * n_out + k];
    a0 += m * x0[q];
    a1 += m * x1[q];
  }
  if (r0 + 2 * g < rows) out[(r0 + 2 * g) * ld_out + k] = a0;
  if (r0 + 2 * g + 1 < rows) out[(r0 + 2 * g + 1) * ld_out + k] = a1;
}

// c[b, f, :] = ([mcep | bap] - mean) / scale as float32 (sklearn StandardScaler.transform, then torch.FloatTensor)
__global__ void pack_features_kernel(const double* __restrict__ a, int na, const double* __restrict__ b, int nb,
                                     const double* __restrict__ mean, const double* __restrict__ scale, float* out,
                                     int64_t rows, int ld_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = na + nb;
  if (i >= rows * nc) return;
  const int64_t r = i / nc;
  const int c = (int)(i - r * nc);
  double v = c < na ? a[r * na + c] : b[r * nb + (c - na)];
  if (mean) v = (v - mean[c]) / scale[c];
  out[r * ld_out + c] = (float)v;
}

// float32 waveform -> the float64 samples the reference's stage reads back from the PCM_16 file the decode CLI wrote
// (libsndfile: lrint(x * 32767) clipped on write, / 32768 on read); pcm16 = 0: plain widening.
__global__ void wave_to_f64_kernel(const float* __restrict__ w, double* __restrict__ out, int64_t n, int pcm16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = (double)w[i];
  if (pcm16) {
    v = rint(v * 32767.0);
    v = fmin(32767.0, fmax(-32768.0, v)) / 32768.0;
  }
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------ F0 contours
// np.interp(np.linspace(0, n_in - 1, n_out), arange(n_in), f0) clamped at 0 (ssc_postprocessing.py:159-166).
// No fused multiply-adds: the fixtures are matched bit for bit.
__global__ void f0_match_length_kernel(const double* __restrict__ in, int64_t in_bs, const int* __restrict__ n_in,
                                       double* __restrict__ out, int64_t out_bs, const int* __restrict__ n_out) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int ni = n_in[b], no = n_out[b];
  if (i >= no) return;
  const double* y = in + (int64_t)b * in_bs;
  double v;
  if (ni == no) {
    v = y[i];  // the reference skips the resampling (and its clamp) for equal lengths
  } else {
    const double stop = (double)(ni - 1);
    const double step = no > 1 ? stop / (double)(no - 1) : 0.0;
    double xv = (double)i * step;
    if (i == no - 1 && no > 1) xv = stop;
    int j = (int)xv;
    if (j >= ni - 1) {
      v = y[ni - 1];
    } else {
      const double slope = (y[j + 1] - y[j]) / 1.0;
      v = slope * (xv - (double)j) + y[j];
    }
    v = fmax(v, 0.0);
  }
  out[(int64_t)b * out_bs + i] = v;
}

// convert_continuos_f0 (ssc_postprocessing.py:51-72): hold the first / last voiced value outwards, join voiced frames
// linearly (scipy interp1d: slope * (x - x_lo) + y_lo with x_lo the last voiced frame strictly before x).
// One workgroup per item; ok[b] = 0 when every frame is unvoiced (cf0 = f0 then).
__global__ __launch_bounds__(NT_F0) void cont_f0_kernel(const double* __restrict__ f0, int64_t bs,
                                                     const int* __restrict__ n_frames, double* __restrict__ cf0,
                                                     float* __restrict__ uv, int* __restrict__ ok) {
#pragma clang fp contract(off)
  __shared__ int s_first, s_last;
  const int b = blockIdx.x;
  const int n = n_frames[b];
  const double* y = f0 + (int64_t)b * bs;
  double* o = cf0 + (int64_t)b * bs;
  float* u = uv + (int64_t)b * bs;
  if (threadIdx.x == 0) {
    s_first = n;
    s_last = -1;
  }
  __syncthreads();
  int first = n, last = -1;
  for (int i = threadIdx.x; i < n; i += NT_F0) {
    const bool v = y[i] != 0.0;
    u[i] = v ? 1.0f : 0.0f;
    if (v) {
      first = min(first, i);
      last = max(last, i);
    }
  }
  atomicMin(&s_first, first);
  atomicMax(&s_last, last);
  __syncthreads();
  first = s_first;
  last = s_last;
  if (last < 0) {
    for (int i = threadIdx.x; i < n; i += NT_F0) o[i] = y[i];
    if (threadIdx.x == 0) ok[b] = 0;
    return;
  }
  if (threadIdx.x == 0) ok[b] = 1;
  // interpolation nodes: every frame up to `first` (holding y[first]), every voiced frame, every frame from `last` on
  auto is_node = [&](int j) { return j <= first || j >= last || y[j] != 0.0; };
  auto value = [&](int j) { return j < first ? y[first] : (j >= last ? y[last] : y[j]); };
  for (int i = threadIdx.x; i < n; i += NT_F0) {
    double v;
    if (n < 2) {
      v = value(i);
    } else {
      int lo = i > 0 ? i - 1 : 0;  // scipy: searchsorted(nodes, i) clipped to [1, len - 1], lo = that - 1
      while (!is_node(lo)) --lo;
      int hi = i > 0 ? i : 1;
      while (!is_node(hi)) ++hi;
      const double y_lo = value(lo), y_hi = value(hi);
      const double slope = (y_hi - y_lo) / (double)(hi - lo);
      v = slope * (double)(i - lo) + y_lo;
    }
    o[i] = v;
  }
}

// exclusive prefix over frames of hop * radius, radius = fl32(fl32(f0) / fs) % 1: SignalGenerator's per-sample cumsum
// restricted to frame starts (the samples of one frame share one radius).  One workgroup per item.
__global__ __launch_bounds__(NT_F0) void sine_phase_kernel(const double* __restrict__ f0, int64_t bs,
                                                        const int* __restrict__ n_frames, int fs, int hop,
                                                        double* __restrict__ phase) {
  __shared__ double red[NT_F0 / 64];
  __shared__ double s_carry;
  const int b = blockIdx.x;
  const int n = n_frames[b];
  const double* y = f0 + (int64_t)b * bs;
  double* o = phase + (int64_t)b * bs;
  if (threadIdx.x == 0) s_carry = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int base = 0; base < n; base += NT_F0) {
    const int i = base + threadIdx.x;
    double v = 0.0;
    if (i < n) {
      const float r = __fdiv_rn((float)y[i], (float)fs);
      v = (double)hop * (double)(r - floorf(r));
    }
    double inc = v;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      const double t = __shfl_up(inc, o2, 64);
      if (lane >= o2) inc += t;
    }
    if (lane == 63) red[w] = inc;
    __syncthreads();
    double run = s_carry + inc - v;
    for (int k = 0; k < w; ++k) run += red[k];
    if (i < n) o[i] = run;
    __syncthreads();
    if (threadIdx.x == NT_F0 - 1) s_carry = run + v;
    __syncthreads();
  }
}

// SignalGenerator(["sine"]) sample by sample + the four dilated-factor tracks, all float32 like the tensors the
// reference hands to SiFiGAN.  grid.x over the n * hop samples of an item, grid.y = item.
struct ExcArgs {
  const double* f0;       // (B, bs) contour driving the sine (cf0)
  const double* df_f0;    // (B, bs) contour driving the dilated factors
  const double* phase;    // (B, bs) from sine_phase_kernel
  const float* noise;     // (B, n_max * hop) standard normal draws or NULL
  float* sine;            // (B, n_max * hop)
  float* dfs[4];          // (B, n_max * us[i])
  int us[4];
  double dense[4];
  int n_df;
};
__global__ void excitation_kernel(ExcArgs a, int64_t bs, const int* __restrict__ n_frames, int n_max, int fs, int hop,
                                  float sine_amp, float noise_amp) {
  const int b = blockIdx.y;
  const int n = n_frames[b];
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= (int64_t)n * hop) return;
  const int f = (int)(s / hop), r = (int)(s - (int64_t)f * hop);
  const double* y = a.f0 + (int64_t)b * bs;
  const float f32 = (float)y[f];
  const float rad = __fdiv_rn(f32, (float)fs);
  const float radius = rad - floorf(rad);
  // torch's CPU cumsum: double accumulation of the float32 radii, every output rounded to float32
  const float cum = (float)(a.phase[(int64_t)b * bs + f] + (double)(r + 1) * (double)radius);
  const float arg = __fmul_rn(__fmul_rn(cum, 2.0f), 3.14159274101257324f);
  const float vuv = f32 > 0.0f ? 1.0f : 0.0f;
  float v = __fmul_rn(__fmul_rn(vuv, (float)sin((double)arg)), sine_amp);
  if (a.noise && noise_amp > 0.0f) {
    const float na = __fadd_rn(__fmul_rn(vuv, noise_amp), __fdiv_rn(__fmul_rn(1.0f - vuv, noise_amp), 3.0f));
    v = __fadd_rn(v, __fmul_rn(a.noise[(int64_t)b * n_max * hop + s], na));
  }
  a.sine[(int64_t)b * n_max * hop + s] = v;
  // dilated factors: every track i holds us[i] samples per frame
  const double* yd = a.df_f0 + (int64_t)b * bs;
  for (int i = 0; i < a.n_df; ++i) {
    if (r < a.us[i]) {
      double d = yd[f];
      if (d == 0.0) d = (double)fs / a.dense[i];
      a.dfs[i][(int64_t)b * n_max * a.us[i] + (int64_t)f * a.us[i] + r] = (float)(1.0 * fs / a.dense[i] / d);
    }
  }
}

int world_lds_bytes(int N, int planes_half, int nt) {  // re, im | quarter-wave twiddles | scratch | planes | reductions
  return (int)sizeof(double) * (2 * N + N / 2 + (N + 8) + planes_half * (N / 2 + 1) + 2 * (nt / 64));
}

int check_world(const SrnWorldParams* p, const char* who) {
  SRN_CHECK_ARG(p != nullptr, "%s: null params", who);
  SRN_CHECK_ARG(p->x && p->x_len && p->f0 && p->t && p->n_frames && p->twiddle, "%s: null input", who);
  SRN_CHECK_ARG(p->n_batch > 0 && p->n_batch <= 65535 && p->max_frames > 0, "%s: bad grid %d x %d", who, p->max_frames,
                p->n_batch);
  SRN_CHECK_ARG(p->fs >= 8000 && p->fs <= 48000, "%s: fs %d outside 8..48 kHz", who, p->fs);
  SRN_CHECK_ARG(p->f_bs >= p->max_frames, "%s: f_bs %lld < max_frames", who, (long long)p->f_bs);
  return 0;
}

}  // namespace

extern "C" int srn_world_cheaptrick(const SrnWorldParams* p, void* stream) {
  if (int rc = check_world(p, "srn_world_cheaptrick")) return rc;
  SRN_CHECK_ARG(p->out0 || p->out1, "srn_world_cheaptrick: no output requested");
  const int half1 = p->fft_size / 2 + 1;
  SRN_CHECK_ARG(!p->out0 || p->ld_out0 >= half1, "srn_world_cheaptrick: ld_out0 < fft_size/2+1");
  SRN_CHECK_ARG(!p->out1 || p->ld_out1 >= half1, "srn_world_cheaptrick: ld_out1 < fft_size/2+1");
  // the window (3 periods of the lowest analysed F0) has to fit the transform
  SRN_CHECK_ARG(p->f0_floor > 0 && 2 * (int)(1.5 * p->fs / p->f0_floor + 0.5) + 1 <= p->fft_size,
                "srn_world_cheaptrick: f0_floor %g too low for fft_size %d", p->f0_floor, p->fft_size);
  dim3 grid(p->max_frames, p->n_batch);
  hipStream_t st = (hipStream_t)stream;
  static SrnSmemAttr a9, a10, a11;
  switch (p->fft_size) {
    case 512: {
      const int lds = world_lds_bytes(512, 1, NT_CHEAPTRICK);
      if (int rc = a9.ensure((const void*)(cheaptrick_kernel<9, NT_CHEAPTRICK>), lds)) return rc;
      hipLaunchKernelGGL((cheaptrick_kernel<9, NT_CHEAPTRICK>), grid, dim3(NT_CHEAPTRICK), lds, st, *p);
      break;
    }
    case 1024: {
      const int lds = world_lds_bytes(1024, 1, NT_CHEAPTRICK);
      if (int rc = a10.ensure((const void*)(cheaptrick_kernel<10, NT_CHEAPTRICK>), lds)) return rc;
      hipLaunchKernelGGL((cheaptrick_kernel<10, NT_CHEAPTRICK>), grid, dim3(NT_CHEAPTRICK), lds, st, *p);
      break;
    }
    case 2048: {
      const int lds = world_lds_bytes(2048, 1, NT_CHEAPTRICK);
      if (int rc = a11.ensure((const void*)(cheaptrick_kernel<11, NT_CHEAPTRICK>), lds)) return rc;
      hipLaunchKernelGGL((cheaptrick_kernel<11, NT_CHEAPTRICK>), grid, dim3(NT_CHEAPTRICK), lds, st, *p);
      break;
    }
    default:
      SRN_CHECK_ARG(false, "srn_world_cheaptrick: fft_size %d (512 / 1024 / 2048)", p->fft_size);
  }
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_world_d4c(const SrnWorldParams* p, void* stream) {
  if (int rc = check_world(p, "srn_world_d4c")) return rc;
  SRN_CHECK_ARG(p->out0 && p->band_window, "srn_world_d4c: null output / band window");
  SRN_CHECK_ARG(p->n_bands >= 1 && p->n_bands <= 8 && p->ld_out0 >= p->n_bands, "srn_world_d4c: n_bands %d", p->n_bands);
  const int N = p->fft_size;
  SRN_CHECK_ARG(N == 1024 || N == 2048, "srn_world_d4c: fft_size %d (1024 / 2048)", N);
  // windows of 4 periods of 47 Hz and of 3 periods of 40 Hz have to fit; so do the bands
  SRN_CHECK_ARG(2 * (int)(2.0 * p->fs / 47.0 + 0.5) + 1 <= N, "srn_world_d4c: fs %d needs a larger transform than %d",
                p->fs, N);
  SRN_CHECK_ARG(p->band_window_len >= 3 && (p->band_window_len & 1) && p->band_window_len <= N,
                "srn_world_d4c: band window length %d", p->band_window_len);
  const int hw = p->band_window_len / 2;
  const int last_center = (int)(3000.0 * p->n_bands * N / p->fs);
  SRN_CHECK_ARG(last_center + hw <= N / 2, "srn_world_d4c: band %d leaves the spectrum", p->n_bands);
  const int boundary = (int)(N * 8.0 / p->band_window_len + 0.5);
  SRN_CHECK_ARG(boundary >= 0 && boundary < N / 2, "srn_world_d4c: boundary %d", boundary);
  static_assert(NT_D4C == 2048 / 4, "d4c_kernel<11> sorts two band powers per thread");
  dim3 grid(p->max_frames, p->n_batch);
  hipStream_t st = (hipStream_t)stream;
  static SrnSmemAttr a10, a11;
  const int nt = N / 4;  // two sorted elements per thread (bitonic_sort_pairs)
  const int lds = world_lds_bytes(N, 2, nt);
  if (N == 1024) {
    if (int rc = a10.ensure((const void*)(d4c_kernel<10, 256>), lds)) return rc;
    hipLaunchKernelGGL((d4c_kernel<10, 256>), grid, dim3(256), lds, st, *p);
  } else {
    if (int rc = a11.ensure((const void*)(d4c_kernel<11, NT_D4C>), lds)) return rc;
    hipLaunchKernelGGL((d4c_kernel<11, NT_D4C>), grid, dim3(NT_D4C), lds, st, *p);
  }
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_world_project(const double* in, int64_t rows, int K, int ld_in, const double* mat_t, int n_out,
                                 int take_log, double* out, int ld_out, void* stream) {
  SRN_CHECK_ARG(in && mat_t && out, "srn_world_project: null pointer");
  SRN_CHECK_ARG(rows > 0 && K > 0 && K <= 2049 && ld_in >= K && n_out > 0 && n_out <= 64 && ld_out >= n_out,
                "srn_world_project: bad shape rows %lld K %d n_out %d", (long long)rows, K, n_out);
  const int lds = PR * K * (int)sizeof(double);
  static SrnSmemAttr attr;
  if (int rc = attr.ensure((const void*)project_kernel, PR * 2049 * (int)sizeof(double))) return rc;
  hipLaunchKernelGGL(project_kernel, dim3((unsigned)((rows + PR - 1) / PR)), dim3(256), lds, (hipStream_t)stream, in,
                     rows, K, ld_in, mat_t, n_out, take_log, out, ld_out);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_world_pack_features(const double* a, int na, const double* b, int nb, const double* mean,
                                       const double* scale, float* out, int64_t rows, int ld_out, void* stream) {
  SRN_CHECK_ARG(a && b && out && na > 0 && nb >= 0 && rows > 0 && ld_out >= na + nb, "srn_world_pack_features: bad args");
  SRN_CHECK_ARG((mean == nullptr) == (scale == nullptr), "srn_world_pack_features: mean and scale go together");
  const int64_t n = rows * (na + nb);
  hipLaunchKernelGGL(pack_features_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, na,
                     b, nb, mean, scale, out, rows, ld_out);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_wave_to_f64(const float* wave, double* out, int64_t n, int pcm16, void* stream) {
  SRN_CHECK_ARG(wave && out && n > 0, "srn_wave_to_f64: bad args");
  hipLaunchKernelGGL(wave_to_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wave, out,
                     n, pcm16);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_f0_match_length(const double* in, int64_t in_bs, const int32_t* n_in, double* out, int64_t out_bs,
                                   const int32_t* n_out, int n_batch, int max_out, void* stream) {
  SRN_CHECK_ARG(in && n_in && out && n_out && n_batch > 0 && n_batch <= 65535 && max_out > 0,
                "srn_f0_match_length: bad args");
  hipLaunchKernelGGL(f0_match_length_kernel, dim3((max_out + 255) / 256, n_batch), dim3(256), 0, (hipStream_t)stream, in,
                     in_bs, n_in, out, out_bs, n_out);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_cont_f0(const double* f0, int64_t bs, const int32_t* n_frames, double* cf0, float* uv, int32_t* ok,
                           int n_batch, void* stream) {
  SRN_CHECK_ARG(f0 && n_frames && cf0 && uv && ok && n_batch > 0, "srn_cont_f0: bad args");
  hipLaunchKernelGGL(cont_f0_kernel, dim3(n_batch), dim3(NT_F0), 0, (hipStream_t)stream, f0, bs, n_frames, cf0, uv, ok);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_sifigan_excitation(const SrnExcitationParams* p, void* stream) {
  SRN_CHECK_ARG(p != nullptr, "srn_sifigan_excitation: null params");
  SRN_CHECK_ARG(p->f0 && p->df_f0 && p->phase_ws && p->sine && p->n_frames, "srn_sifigan_excitation: null pointer");
  SRN_CHECK_ARG(p->n_batch > 0 && p->n_batch <= 65535 && p->max_frames > 0 && p->hop > 0 && p->fs > 0,
                "srn_sifigan_excitation: bad sizes");
  SRN_CHECK_ARG(p->n_df >= 0 && p->n_df <= 4, "srn_sifigan_excitation: n_df %d", p->n_df);
  ExcArgs a;
  a.f0 = p->f0;
  a.df_f0 = p->df_f0;
  a.phase = p->phase_ws;
  a.noise = p->noise;
  a.sine = p->sine;
  a.n_df = p->n_df;
  for (int i = 0; i < 4; ++i) {
    a.dfs[i] = i < p->n_df ? p->dfs[i] : nullptr;
    a.us[i] = i < p->n_df ? p->df_upsample[i] : 0;
    a.dense[i] = i < p->n_df ? p->dense_factors[i] : 1.0;
    if (i < p->n_df)
      SRN_CHECK_ARG(a.dfs[i] && a.us[i] > 0 && a.us[i] <= p->hop && a.dense[i] > 0,
                    "srn_sifigan_excitation: dilated-factor track %d", i);
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sine_phase_kernel, dim3(p->n_batch), dim3(NT_F0), 0, st, p->f0, p->f_bs, p->n_frames, p->fs, p->hop,
                     p->phase_ws);
  const int64_t n = (int64_t)p->max_frames * p->hop;
  hipLaunchKernelGGL(excitation_kernel, dim3((unsigned)((n + 255) / 256), p->n_batch), dim3(256), 0, st, a, p->f_bs,
                     p->n_frames, p->max_frames, p->fs, p->hop, p->sine_amp, p->noise_amp);
  SRN_CHECK_LAUNCH();
  return 0;
}
