// train.hip -- backward and optimizer kernels of the estimator's training step (the backward pass of
// the blocks of matcha_components/decoder.py and transformer.py, reached from CFM.compute_loss,
// flow_matching.py:95-133, and the AdamW update of bin/ssc_train.py:331-349 / trainers/ssc.py:86-96).
//
// All of these are HBM-bound row / column reductions and elementwise maps over channels-last fp32 (B, T, C); the
// GEMM-shaped gradients (dgrad of every conv / projection, dP and dQ of attention) go through srn_conv_gemm.
//   srn_rowln_fwd / srn_rowln_bwd   per-frame LayerNorm over C with a per-(batch, channel) multiplier and offset:
//                                   nn.LayerNorm (batch stride 0) and SpeakerAdapter (decoder.py:34-45)
//   srn_gn_mish_bwd_partial / _apply  GroupNorm(8) -> Mish -> mask of Block1D (decoder.py:66-77), statistics over
//                                   the padded length like the reference
//   srn_softmax_bwd                 dS = scale * P o (dP - rowsum(dP o P)), in place
//   srn_geglu_fwd / srn_geglu_bwd   h * gelu_erf(g) (transformer.py:120-146) with both halves kept for backward
//   srn_adamw                       torch.optim.AdamW step on one flat fp32 buffer, gradient scale (clip) folded in
// Column sums (d gamma, d beta, per-batch adapter gradients) leave as per-row-chunk partial sums [..][chunk][2][C];
// the host adds the few chunks (no atomics: results are bit-reproducible).
#include <hip/hip_runtime.h>

#include "common.h"
#include "serenade_hip.h"

namespace {

constexpr int MAXV = 4;       // float4 per lane: C <= 1024
constexpr int LN_ROWS = 8;    // rows per workgroup of the row-LayerNorm backward (one chunk of partial sums): 32 left a
                              // B = 4 x L = 1024 step with 128 workgroups on 256 CUs (24 us per launch)
constexpr int GN_ROWS = 8;    // rows per workgroup of the GroupNorm backward reduction (same reason)

// d/dx [x tanh(softplus(x))] = th + x (1 - th^2) sigmoid(x), th = tanh(softplus(x)) = n / (n + 2), n = e^x (e^x + 2)
__device__ __forceinline__ float mish_grad(float x) {
  const float e = expf(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  const float th = n / (n + 2.0f);
  const float sg = 1.0f / (1.0f + expf(-x));
  const float g = th + x * (1.0f - th * th) * sg;
  return x > 20.0f ? 1.0f : g;
}

// d/dx gelu_erf(x) = Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// ---- per-frame LayerNorm with per-(batch, channel) multiplier m and offset a: y = xhat * m[b] + a[b]
__global__ __launch_bounds__(256) void rowln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                        int64_t m_bs, const float* __restrict__ a, int64_t a_bs,
                                                        float* __restrict__ y, int T, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int c4n = C / 4;
  const float inv_c = 1.0f / (float)C;
  m += (int64_t)b * m_bs;
  a += (int64_t)b * a_bs;
  for (int t = blockIdx.x * 4 + wave; t < T; t += gridDim.x * 4) {
    const int64_t row = ((int64_t)b * T + t) * C;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < c4n) {
        v[i] = *reinterpret_cast<const float4*>(x + row + c4 * 4);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < c4n) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_c + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < c4n) {
        const int c = c4 * 4;
        const float4 mm = *reinterpret_cast<const float4*>(m + c);
        const float4 aa = *reinterpret_cast<const float4*>(a + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * mm.x + aa.x;
        o.y = (v[i].y - mean) * rstd * mm.y + aa.y;
        o.z = (v[i].z - mean) * rstd * mm.z + aa.z;
        o.w = (v[i].w - mean) * rstd * mm.w + aa.w;
        *reinterpret_cast<float4*>(y + row + c) = o;
      }
    }
  }
}

// dx = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = dy * m[b]; partial[b][chunk][0][c] = sum_t dy xhat,
// partial[b][chunk][1][c] = sum_t dy over the chunk's LN_ROWS rows.  One workgroup per (chunk, b); a wave per row.
__global__ __launch_bounds__(256) void rowln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ m, int64_t m_bs,
                                                        float* __restrict__ dx, float* __restrict__ partial, int T,
                                                        int C, float eps) {
  __shared__ float red[2][4][MAXV * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y, chunk = blockIdx.x, n_chunk = gridDim.x;
  const int c4n = C / 4;
  const float inv_c = 1.0f / (float)C;
  m += (int64_t)b * m_bs;
  float4 am[MAXV], aa[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) am[i] = aa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int t_end = min(T, (chunk + 1) * LN_ROWS);
  for (int t = chunk * LN_ROWS + wave; t < t_end; t += 4) {
    const int64_t row = ((int64_t)b * T + t) * C;
    float4 v[MAXV], g[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      v[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < c4n) {
        v[i] = *reinterpret_cast<const float4*>(x + row + c4 * 4);
        g[i] = *reinterpret_cast<const float4*>(dy + row + c4 * 4);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < c4n) {
        const float dx0 = v[i].x - mean, dx1 = v[i].y - mean, dx2 = v[i].z - mean, dx3 = v[i].w - mean;
        sq += (dx0 * dx0 + dx1 * dx1) + (dx2 * dx2 + dx3 * dx3);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
    float4 gh[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      gh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < c4n) {
        const float4 mm = *reinterpret_cast<const float4*>(m + c4 * 4);
        v[i].x = (v[i].x - mean) * rstd, v[i].y = (v[i].y - mean) * rstd;  // xhat
        v[i].z = (v[i].z - mean) * rstd, v[i].w = (v[i].w - mean) * rstd;
        am[i].x += g[i].x * v[i].x, am[i].y += g[i].y * v[i].y, am[i].z += g[i].z * v[i].z, am[i].w += g[i].w * v[i].w;
        aa[i].x += g[i].x, aa[i].y += g[i].y, aa[i].z += g[i].z, aa[i].w += g[i].w;
        gh[i].x = g[i].x * mm.x, gh[i].y = g[i].y * mm.y, gh[i].z = g[i].z * mm.z, gh[i].w = g[i].w * mm.w;
        s1 += (gh[i].x + gh[i].y) + (gh[i].z + gh[i].w);
        s2 += (gh[i].x * v[i].x + gh[i].y * v[i].y) + (gh[i].z * v[i].z + gh[i].w * v[i].w);
      }
    }
    s1 = wave_sum(s1) * inv_c;
    s2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < c4n) {
        float4 o;
        o.x = rstd * (gh[i].x - s1 - v[i].x * s2);
        o.y = rstd * (gh[i].y - s1 - v[i].y * s2);
        o.z = rstd * (gh[i].z - s1 - v[i].z * s2);
        o.w = rstd * (gh[i].w - s1 - v[i].w * s2);
        *reinterpret_cast<float4*>(dx + row + c4 * 4) = o;
      }
    }
  }
  // the four waves' column sums -> one partial row pair
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      *reinterpret_cast<float4*>(&red[0][wave][c]) = am[i];
      *reinterpret_cast<float4*>(&red[1][wave][c]) = aa[i];
    }
  }
  __syncthreads();
  float* out = partial + ((int64_t)b * n_chunk + chunk) * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int k = c / C, cc = c - k * C;
    out[c] = (red[k][0][cc] + red[k][1][cc]) + (red[k][2][cc] + red[k][3][cc]);
  }
}

// ---- GroupNorm -> Mish -> mask, backward.  g = gamma xhat + beta, y = mish(g) [t < len], dg = dy mish'(g).
// partial[b][chunk][0][c] = sum_t dg, [1][c] = sum_t dg xhat  over the chunk's rows (all T rows: padded rows give 0).
__global__ __launch_bounds__(256) void gn_mish_bwd_partial_kernel(
    const float* __restrict__ h, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const int32_t* __restrict__ lens, float* __restrict__ partial, int T, int C, int groups) {
  __shared__ float red[2][256 * 4];
  const int b = blockIdx.y, chunk = blockIdx.x, n_chunk = gridDim.x;
  const int c4n = C / 4;           // <= 256
  const int rsub_n = 256 / c4n;    // row lanes (>= 1)
  const int c4 = threadIdx.x % c4n, rsub = threadIdx.x / c4n;
  const int len = lens ? min(lens[b], T) : T;
  const int cpg = C / groups;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (rsub < rsub_n) {
    const int c = c4 * 4;
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    const float4 be = *reinterpret_cast<const float4*>(beta + c);
    const int g = c / cpg;
    const float mu = mean[b * groups + g], rs = rstd[b * groups + g];
    const int t_end = min(min(T, (chunk + 1) * GN_ROWS), len);
    for (int t = chunk * GN_ROWS + rsub; t < t_end; t += rsub_n) {
      const int64_t at = ((int64_t)b * T + t) * C + c;
      const float4 hv = *reinterpret_cast<const float4*>(h + at);
      const float4 gv = *reinterpret_cast<const float4*>(dy + at);
      float4 xh, dg;
      xh.x = (hv.x - mu) * rs, xh.y = (hv.y - mu) * rs, xh.z = (hv.z - mu) * rs, xh.w = (hv.w - mu) * rs;
      dg.x = gv.x * mish_grad(xh.x * ga.x + be.x);
      dg.y = gv.y * mish_grad(xh.y * ga.y + be.y);
      dg.z = gv.z * mish_grad(xh.z * ga.z + be.z);
      dg.w = gv.w * mish_grad(xh.w * ga.w + be.w);
      a0.x += dg.x, a0.y += dg.y, a0.z += dg.z, a0.w += dg.w;
      a1.x += dg.x * xh.x, a1.y += dg.y * xh.y, a1.z += dg.z * xh.z, a1.w += dg.w * xh.w;
    }
  }
  *reinterpret_cast<float4*>(&red[0][threadIdx.x * 4]) = a0;
  *reinterpret_cast<float4*>(&red[1][threadIdx.x * 4]) = a1;
  __syncthreads();
  float* out = partial + ((int64_t)b * n_chunk + chunk) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float s = 0.f;
    for (int r = 0; r < rsub_n; ++r) s += red[k][(r * c4n + c / 4) * 4 + (c & 3)];
    out[i] = s;
  }
}

// dh = rstd (dg gamma - A / n - xhat Bq / n), (A, Bq)[b][g] = sum over the group of (dg gamma, dg gamma xhat), n = T C/G
__global__ __launch_bounds__(256) void gn_mish_bwd_apply_kernel(
    const float* __restrict__ h, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ gsum, const int32_t* __restrict__ lens, float* __restrict__ dh, int T, int C, int groups) {
  const int b = blockIdx.y;
  const int c4n = C / 4;
  const int len = lens ? min(lens[b], T) : T;
  const int cpg = C / groups;
  const float inv_n = 1.0f / ((float)T * (float)cpg);
  const int r0 = blockIdx.x * 8;
  const int total = min(8, T - r0) * c4n;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int r = idx / c4n;
    const int c = (idx - r * c4n) * 4;
    const int t = r0 + r;
    const int g = c / cpg;
    const float mu = mean[b * groups + g], rs = rstd[b * groups + g];
    const float A = gsum[(b * groups + g) * 2] * inv_n, Bq = gsum[(b * groups + g) * 2 + 1] * inv_n;
    const int64_t at = ((int64_t)b * T + t) * C + c;
    const float4 hv = *reinterpret_cast<const float4*>(h + at);
    float4 xh, dg = make_float4(0.f, 0.f, 0.f, 0.f);
    xh.x = (hv.x - mu) * rs, xh.y = (hv.y - mu) * rs, xh.z = (hv.z - mu) * rs, xh.w = (hv.w - mu) * rs;
    if (t < len) {
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
      const float4 be = *reinterpret_cast<const float4*>(beta + c);
      const float4 gv = *reinterpret_cast<const float4*>(dy + at);
      dg.x = gv.x * mish_grad(xh.x * ga.x + be.x) * ga.x;
      dg.y = gv.y * mish_grad(xh.y * ga.y + be.y) * ga.y;
      dg.z = gv.z * mish_grad(xh.z * ga.z + be.z) * ga.z;
      dg.w = gv.w * mish_grad(xh.w * ga.w + be.w) * ga.w;
    }
    float4 o;
    o.x = rs * (dg.x - A - xh.x * Bq);
    o.y = rs * (dg.y - A - xh.y * Bq);
    o.z = rs * (dg.z - A - xh.z * Bq);
    o.w = rs * (dg.w - A - xh.w * Bq);
    *reinterpret_cast<float4*>(dh + at) = o;
  }
}

// ---- GroupNorm statistics from the conv epilogue's 32 x 32 tile sums: (mean, rstd)[b][g], fp64 accumulation,
// statistics over the padded length T (the arithmetic of norm_act.hip's group_stats, kept for the backward pass)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ partials, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int T, int C, int groups, float eps) {
  const int b = blockIdx.x;
  const int gn_mt = (T + 31) / 32, gn_nt = C / 32, nt_per_g = (C / groups) / 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = partials + (int64_t)b * gn_mt * gn_nt * 2;
  const int per_g = gn_mt * nt_per_g;
  for (int g = wave; g < groups; g += 4) {
    double s1 = 0.0, s2 = 0.0;
    for (int e = lane; e < per_g; e += 64) {
      const int mt = e / nt_per_g;
      const int nt = g * nt_per_g + (e - mt * nt_per_g);
      const float2 v = *reinterpret_cast<const float2*>(base + ((int64_t)mt * gn_nt + nt) * 2);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane == 0) {
      const double cnt = (double)T * (double)(C / groups);
      const double m = s1 / cnt;
      double var = s2 / cnt - m * m;
      if (var < 0.0) var = 0.0;
      mean[b * groups + g] = (float)m;
      rstd[b * groups + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
}

// ---- column sums of per-chunk partial sums: col[b][i] = sum_chunk partial[b][chunk][i], i < n_out (= 2 C for the
// (.., 2, C) buffers of the kernels above), one output per thread; and (optional) the per-group sums of gamma * col that
// the GroupNorm backward needs, gsum[b][g][k], summed in a fixed order by the block that owns the group's channels
__global__ __launch_bounds__(256) void chunk_colsum_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                           float* __restrict__ col, float* __restrict__ gsum, int n_chunk,
                                                           int n_out, int C, int groups) {
  __shared__ float cs[256];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* p = partial + (int64_t)b * n_chunk * n_out;
  float s = 0.f;
  if (i < n_out) {
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains keep the loads in flight
    int ch = 0;
    for (; ch + 4 <= n_chunk; ch += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] += p[(int64_t)(ch + u) * n_out + i];
    }
    for (; ch < n_chunk; ++ch) acc4[0] += p[(int64_t)ch * n_out + i];
    s = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
    col[(int64_t)b * n_out + i] = s;
  }
  if (gsum == nullptr) return;
  // host guarantees C % 256 == 0 and 256 % (C / groups) == 0: a block's 256 outputs are whole groups of one k
  const int k = (blockIdx.x * 256) / C, c0 = blockIdx.x * 256 - k * C;
  cs[threadIdx.x] = i < n_out ? s * gamma[c0 + threadIdx.x] : 0.f;
  __syncthreads();
  const int cpg = C / groups;
  if (threadIdx.x < 256 / cpg) {
    float t = 0.f;
    for (int c = 0; c < cpg; ++c) t += cs[threadIdx.x * cpg + c];
    gsum[((int64_t)b * groups + c0 / cpg + threadIdx.x) * 2 + k] = t;
  }
}

// ---- column sums of a row-major (R, N) matrix, stage 1: partial[chunk][n] = sum of COLSUM_ROWS rows (coalesced over n);
// stage 2 is chunk_colsum_kernel.  The bias gradients and the broadcast-add gradients of the training step.
constexpr int COLSUM_ROWS = 32;  // small chunks: (N / 256) x (R / 32) workgroups keep the chip busy on an 8 MB matrix
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                             int64_t R, int N, int ld) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM_ROWS;
  x += (int64_t)blockIdx.z * R * ld;                       // batch item
  partial += (int64_t)blockIdx.z * gridDim.y * N;
  if (n >= N) return;
  const int64_t r1 = min(R, r0 + COLSUM_ROWS);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t r = r0;
  for (; r + 4 <= r1; r += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += x[(r + u) * ld + n];
  }
  for (; r < r1; ++r) a[0] += x[r * ld + n];
  partial[(int64_t)blockIdx.y * N + n] = (a[0] + a[1]) + (a[2] + a[3]);
}

// ---- softmax backward in place: dp <- scale * p o (dp - sum_j dp_j p_j); one wave per row of L (row stride ld)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                          int64_t rows, int L, int ld, float scale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float* pr = p + r * ld;
  float* dr = dp + r * ld;
  const int l4 = L / 4;  // L % 4 == 0 is not required: tail handled scalar
  float s = 0.f;
  for (int i = lane; i < l4; i += 64) {
    const float4 a = *reinterpret_cast<const float4*>(pr + i * 4);
    const float4 g = *reinterpret_cast<const float4*>(dr + i * 4);
    s += (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
  }
  for (int j = l4 * 4 + lane; j < L; j += 64) s += pr[j] * dr[j];
  s = wave_sum(s);
  for (int i = lane; i < l4; i += 64) {
    const float4 a = *reinterpret_cast<const float4*>(pr + i * 4);
    float4 g = *reinterpret_cast<const float4*>(dr + i * 4);
    g.x = scale * a.x * (g.x - s), g.y = scale * a.y * (g.y - s);
    g.z = scale * a.z * (g.z - s), g.w = scale * a.w * (g.w - s);
    *reinterpret_cast<float4*>(dr + i * 4) = g;
  }
  for (int j = l4 * 4 + lane; j < L; j += 64) dr[j] = scale * pr[j] * (dr[j] - s);
}

// ---- GEGLU: hg (rows, 2 inner) = [h | g]; a = h gelu(g)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const float* __restrict__ hg, float* __restrict__ a,
                                                        int64_t rows, int inner) {
  const int64_t n4 = rows * (inner / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (inner / 4);
    const int c = (int)(i - r * (inner / 4)) * 4;
    const float4 h = *reinterpret_cast<const float4*>(hg + r * 2 * inner + c);
    const float4 g = *reinterpret_cast<const float4*>(hg + r * 2 * inner + inner + c);
    float4 o;
    o.x = h.x * srn_gelu_erf(g.x), o.y = h.y * srn_gelu_erf(g.y);
    o.z = h.z * srn_gelu_erf(g.z), o.w = h.w * srn_gelu_erf(g.w);
    *reinterpret_cast<float4*>(a + r * inner + c) = o;
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ hg, const float* __restrict__ da,
                                                        float* __restrict__ dhg, int64_t rows, int inner) {
  const int64_t n4 = rows * (inner / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (inner / 4);
    const int c = (int)(i - r * (inner / 4)) * 4;
    const float4 h = *reinterpret_cast<const float4*>(hg + r * 2 * inner + c);
    const float4 g = *reinterpret_cast<const float4*>(hg + r * 2 * inner + inner + c);
    const float4 d = *reinterpret_cast<const float4*>(da + r * inner + c);
    float4 dh, dg;
    dh.x = d.x * srn_gelu_erf(g.x), dh.y = d.y * srn_gelu_erf(g.y);
    dh.z = d.z * srn_gelu_erf(g.z), dh.w = d.w * srn_gelu_erf(g.w);
    dg.x = d.x * h.x * gelu_grad(g.x), dg.y = d.y * h.y * gelu_grad(g.y);
    dg.z = d.z * h.z * gelu_grad(g.z), dg.w = d.w * h.w * gelu_grad(g.w);
    *reinterpret_cast<float4*>(dhg + r * 2 * inner + c) = dh;
    *reinterpret_cast<float4*>(dhg + r * 2 * inner + inner + c) = dg;
  }
}

// ---- torch.optim.AdamW (decoupled weight decay) on a flat buffer; g is multiplied by gscale first (gradient clipping)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    float gscale) {
  const float step = lr / bc1;
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi, m[i] = mi, v[i] = vi;
  }
}

// the same update with its step-dependent scalars read from device memory (dyn = {lr, bc1, bc2, grad_scale}): the
// launch is then identical every step and can sit in a captured hipGraph
__global__ __launch_bounds__(256) void adamw_dyn_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float beta1, float beta2, float eps, float wd,
                                                        const float* __restrict__ dyn) {
  const float lr = dyn[0], bc1 = dyn[1], bc2 = dyn[2], gscale = dyn[3];
  const float step = lr / bc1;
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    p[i] = pi, m[i] = mi, v[i] = vi;
  }
}

// sum of squares in fp64: partial[block] for <= 1024 blocks (the host adds them): clip_grad_norm_'s total norm
// (h == g: sum of squares; h == nullptr: plain sum; else the dot product -- the loss sums of the training step)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const float* __restrict__ h, int64_t n,
                                                    double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(g + i * 4);
    const float4 w = h ? *reinterpret_cast<const float4*>(h + i * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    acc += ((double)v.x * w.x + (double)v.y * w.y) + ((double)v.z * w.z + (double)v.w * w.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)g[i] * (h ? h[i] : 1.f);
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- many small device-to-device copies in one launch: entry e copies len[e] floats from src[e] to dst + off[e].
// The table travels in the kernel arguments (no host-to-device copy, so a captured step can contain it).
__global__ __launch_bounds__(256) void multi_copy_kernel(const SrnCopyList list, float* __restrict__ dst) {
  const int e = blockIdx.x;
  const int64_t n = list.len[e];
  const float* __restrict__ src = reinterpret_cast<const float*>(list.src[e]);
  float* __restrict__ d = dst + list.off[e];
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(d)) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.y * 256)
    reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(src)[i];
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.y * 256) d[i] = src[i];
}

// ---- weight norm + re-layout, one workgroup per output channel (see serenade_hip.h)
__device__ inline float block_sum_256(float x, float* red) {  // fixed order: lanes, then the four waves 0..3
  x = wave_sum(x);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  const float r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ w, float* __restrict__ wd,
                                                              float* __restrict__ inv_norm, int N, int C, int k) {
  __shared__ float red[4];
  const int n = blockIdx.x, E = C * k;
  const float* vn = v + (int64_t)n * E;
  float ss = 0.f;
  for (int e = threadIdx.x; e < E; e += 256) ss += vn[e] * vn[e];
  ss = block_sum_256(ss, red);
  const float inv = 1.0f / sqrtf(ss);
  const float sc = g[n] * inv;
  if (threadIdx.x == 0) inv_norm[n] = inv;
  float* wn = w + (int64_t)n * E;
  for (int e = threadIdx.x; e < E; e += 256) {  // e walks the PACKED row: (j, c)
    const int j = e / C, c = e - j * C;
    const float x = sc * vn[c * k + j];
    wn[e] = x;
    if (wd) wd[(int64_t)c * k * N + (int64_t)j * N + n] = x;
  }
}

__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                              const float* __restrict__ g,
                                                              const float* __restrict__ inv_norm, float* __restrict__ dv,
                                                              float* __restrict__ dg, int N, int C, int k) {
  __shared__ float red[4];
  const int n = blockIdx.x, E = C * k;
  const float* vn = v + (int64_t)n * E;
  const float* dn = dw + (int64_t)n * E;
  float dot = 0.f;
  for (int e = threadIdx.x; e < E; e += 256) {  // e walks v's row: (c, j)
    const int c = e / k, j = e - c * k;
    dot += dn[j * C + c] * vn[e];
  }
  dot = block_sum_256(dot, red);
  const float inv = inv_norm[n], gn = g[n];
  if (threadIdx.x == 0) dg[n] = dot * inv;
  const float a = gn * inv, b = dot * inv * inv;
  float* o = dv + (int64_t)n * E;
  for (int e = threadIdx.x; e < E; e += 256) {
    const int c = e / k, j = e - c * k;
    o[e] = a * (dn[j * C + c] - vn[e] * b);
  }
}

inline unsigned grid_for(int64_t n, int64_t cap = 16384) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int srn_weight_norm_fwd(const float* v, const float* g, float* w_packed, float* wd, float* inv_norm, int N,
                                   int C, int k, void* stream) {
  SRN_CHECK_ARG(v && g && w_packed && inv_norm && N > 0 && C > 0 && k > 0, "weight_norm_fwd: bad args");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, v, g, w_packed, wd,
                     inv_norm, N, C, k);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_weight_norm_bwd(const float* dw_packed, const float* v, const float* g, const float* inv_norm,
                                   float* dv, float* dg, int N, int C, int k, void* stream) {
  SRN_CHECK_ARG(dw_packed && v && g && inv_norm && dv && dg && N > 0 && C > 0 && k > 0, "weight_norm_bwd: bad args");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, dw_packed, v, g,
                     inv_norm, dv, dg, N, C, k);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_rowln_fwd(const float* x, const float* m, int64_t m_bs, const float* a, int64_t a_bs, float* y, int B,
                             int T, int C, float eps, void* stream) {
  SRN_CHECK_ARG(x && m && a && y && B > 0 && T > 0, "rowln_fwd: bad args");
  SRN_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * MAXV, "rowln_fwd: C=%d unsupported", C);
  int blocks = (T + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(rowln_fwd_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, m, m_bs,
                     a, a_bs, y, T, C, eps);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_rowln_chunks(int T) { return (T + LN_ROWS - 1) / LN_ROWS; }
extern "C" int srn_gn_chunks(int T) { return (T + GN_ROWS - 1) / GN_ROWS; }

extern "C" int srn_rowln_bwd(const float* x, const float* dy, const float* m, int64_t m_bs, float* dx, float* partial,
                             int B, int T, int C, float eps, void* stream) {
  SRN_CHECK_ARG(x && dy && m && dx && partial && B > 0 && T > 0, "rowln_bwd: bad args");
  SRN_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 256 * MAXV, "rowln_bwd: C=%d unsupported", C);
  hipLaunchKernelGGL(rowln_bwd_kernel, dim3((unsigned)srn_rowln_chunks(T), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, dy, m, m_bs, dx, partial, T, C, eps);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gn_mish_bwd_partial(const float* h, const float* dy, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, const int32_t* lens, float* partial,
                                       int B, int T, int C, int groups, void* stream) {
  SRN_CHECK_ARG(h && dy && mean && rstd && gamma && beta && partial && B > 0 && T > 0, "gn_mish_bwd_partial: bad args");
  SRN_CHECK_ARG(C > 0 && C % 4 == 0 && C / 4 <= 256 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0,
                "gn_mish_bwd_partial: C=%d groups=%d unsupported", C, groups);
  hipLaunchKernelGGL(gn_mish_bwd_partial_kernel, dim3((unsigned)srn_gn_chunks(T), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, h, dy, mean, rstd, gamma, beta, lens, partial, T, C, groups);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gn_mish_bwd_apply(const float* h, const float* dy, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, const float* gsum, const int32_t* lens,
                                     float* dh, int B, int T, int C, int groups, void* stream) {
  SRN_CHECK_ARG(h && dy && mean && rstd && gamma && beta && gsum && dh && B > 0 && T > 0, "gn_mish_bwd_apply: bad args");
  SRN_CHECK_ARG(C > 0 && C % 4 == 0 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0,
                "gn_mish_bwd_apply: C=%d groups=%d unsupported", C, groups);
  hipLaunchKernelGGL(gn_mish_bwd_apply_kernel, dim3((unsigned)((T + 7) / 8), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, h, dy, mean, rstd, gamma, beta, gsum, lens, dh, T, C, groups);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_softmax_bwd(const float* p, float* dp, int64_t rows, int L, int ld, float scale, void* stream) {
  SRN_CHECK_ARG(p && dp && rows > 0 && L > 0 && ld >= L && ld % 4 == 0, "softmax_bwd: bad args");
  const int64_t blocks = (rows + 3) / 4;
  SRN_CHECK_ARG(blocks < (1ll << 31), "softmax_bwd: too many rows");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, dp, rows, L, ld,
                     scale);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_geglu_fwd(const float* hg, float* a, int64_t rows, int inner, void* stream) {
  SRN_CHECK_ARG(hg && a && rows > 0 && inner > 0 && inner % 4 == 0, "geglu_fwd: bad args");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * (inner / 4))), dim3(256), 0, (hipStream_t)stream, hg, a, rows,
                     inner);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_geglu_bwd(const float* hg, const float* da, float* dhg, int64_t rows, int inner, void* stream) {
  SRN_CHECK_ARG(hg && da && dhg && rows > 0 && inner > 0 && inner % 4 == 0, "geglu_bwd: bad args");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * (inner / 4))), dim3(256), 0, (hipStream_t)stream, hg, da, dhg,
                     rows, inner);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, void* stream) {
  SRN_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "adamw: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2, grad_scale);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_adamw_dyn(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                             float weight_decay, const float* dyn, void* stream) {
  SRN_CHECK_ARG(p && g && m && v && dyn && n > 0, "adamw_dyn: bad args");
  hipLaunchKernelGGL(adamw_dyn_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                     eps, weight_decay, dyn);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_sumsq_blocks(int64_t n) { return (int)grid_for((n + 7) / 8, 1024); }

extern "C" int srn_sumsq(const float* g, int64_t n, double* partial, void* stream) {
  SRN_CHECK_ARG(g && partial && n > 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, "sumsq: bad args");
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)srn_sumsq_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, g, n, partial);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_dot(const float* a, const float* b, int64_t n, double* partial, void* stream) {
  SRN_CHECK_ARG(a && partial && n > 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(b) & 15) == 0, "dot: bad args");
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)srn_sumsq_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, partial);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gn_stats(const float* partials, float* mean, float* rstd, int B, int T, int C, int groups, float eps,
                            void* stream) {
  SRN_CHECK_ARG(partials && mean && rstd && B > 0 && T > 0, "gn_stats: bad args");
  SRN_CHECK_ARG(C > 0 && groups > 0 && C % groups == 0 && (C / groups) % 32 == 0, "gn_stats: C=%d groups=%d unsupported", C,
                groups);
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, partials, mean, rstd, T, C,
                     groups, eps);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_chunk_colsum(const float* partial, const float* gamma, float* col, float* gsum, int B, int n_chunk, int C,
                                int groups, void* stream) {
  SRN_CHECK_ARG(partial && col && B > 0 && n_chunk > 0 && C > 0, "chunk_colsum: bad args");
  if (gsum != nullptr)
    SRN_CHECK_ARG(gamma != nullptr && groups > 0 && C % groups == 0 && C % 256 == 0 && 256 % (C / groups) == 0,
                  "chunk_colsum: C=%d groups=%d unsupported with gsum", C, groups);
  const int n_out = 2 * C;
  hipLaunchKernelGGL(chunk_colsum_kernel, dim3((unsigned)((n_out + 255) / 256), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, partial, gamma, col, gsum, n_chunk, n_out, C, groups);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_colsum_chunks(int64_t R) { return (int)((R + COLSUM_ROWS - 1) / COLSUM_ROWS); }

extern "C" int srn_colsum(const float* x, float* partial, float* out, int B, int64_t R, int N, int ld, void* stream) {
  SRN_CHECK_ARG(x && partial && out && B > 0 && B < 65536 && R > 0 && N > 0 && ld >= N, "colsum: bad args");
  const int chunks = srn_colsum_chunks(R);
  SRN_CHECK_ARG(chunks < 65536, "colsum: too many rows");
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)chunks, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, x, partial, R, N, ld);
  SRN_CHECK_LAUNCH();
  hipLaunchKernelGGL(chunk_colsum_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partial, (const float*)nullptr, out, (float*)nullptr, chunks, N, N, 1);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_multi_copy(const SrnCopyList* list, float* dst, void* stream) {
  SRN_CHECK_ARG(list && dst && list->n > 0 && list->n <= SRN_COPY_LIST_MAX, "multi_copy: bad list");
  int64_t longest = 0;
  for (int e = 0; e < list->n; ++e) {
    SRN_CHECK_ARG(list->src[e] != nullptr && list->len[e] >= 0 && list->off[e] >= 0, "multi_copy: bad entry %d", e);
    longest = list->len[e] > longest ? list->len[e] : longest;
  }
  int64_t by = (longest / 4 + 255) / 256 / 4;  // ~4 float4 per thread on the longest entry
  by = by < 1 ? 1 : (by > 256 ? 256 : by);
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)list->n, (unsigned)by), dim3(256), 0, (hipStream_t)stream, *list,
                     dst);
  SRN_CHECK_LAUNCH();
  return 0;
}
