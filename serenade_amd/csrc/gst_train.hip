// gst_train.hip -- training-mode kernels of the GST style encoder (the reference runs
// serenade/modules/gst/style_encoder.py:171-191 (ReferenceEncoder: Conv2d + BatchNorm2d + ReLU x 6, GRU) and :235-252
// (StyleTokenLayer) under autograd in trainers/ssc.py:57-96).  The convolutions themselves are srn_conv_gemm /
// srn_tn_gemm launches (one kernel row at a time, see training.py); this file holds what is left:
//   * BatchNorm2d in training mode + ReLU on channels-last rows, forward and backward (per-channel batch statistics
//     as fixed-order chunked column sums: no atomics, bit-reproducible);
//   * the GRU recurrence over T_ref / 64 steps on a precomputed input projection, forward (keeping the gates) and
//     back-propagation through time;
//   * the 4-head attention of one query per utterance over the 50 style tokens, forward and backward.
// All of it is latency-bound (a few thousand elements per utterance); the point is that no library (MIOpen, rocBLAS)
// and no chain of ~300 tiny tensor ops is left in the step.
#include <hip/hip_runtime.h>

#include "common.h"
#include "serenade_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ BatchNorm + ReLU
// rows per partial-sum chunk: at most ~256 chunks, so the finishing kernel's ordered sum stays short
__host__ __device__ inline int64_t bn_rows_per_chunk(int64_t R) { return R <= 256 * 16 ? 16 : (R + 255) / 256; }
// (bn_partial_kernel: a thread owns FOUR channels (16-B loads) of up to 1024 per workgroup; with fewer channels -- the GST's
// 32 .. 128 -- the spare threads take further rows of the chunk (256 / (C / 4) row lanes), joined in lane order through
// LDS; bn_finish_kernel adds the chunks)

// mode 0: partial[chunk][0][c] = sum x, [1][c] = sum x^2
// mode 1: g = dy * (y > 0): [0][c] = sum g, [1][c] = sum g * xhat,  xhat = (x - mean) * rstd
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dy, const float* __restrict__ stats,
                                                         float* __restrict__ partial, int64_t R, int C, int mode) {
  __shared__ float4 red[2][256];
  const int c4n = C / 4;                      // C % 4 == 0 (checked by the callers)
  const int cw = c4n < 256 ? c4n : 256;       // channel quads per workgroup
  const int lanes = 256 / cw;                 // row lanes per quad (>= 1); threads past lanes * cw idle
  const int lc = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int c = (blockIdx.x * cw + lc) * 4;
  const bool live = rl < lanes && c < C;
  const int64_t per = bn_rows_per_chunk(R);
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(R, r0 + per);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (live) {
    if (mode == 0) {
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
        s0.x += v.x, s0.y += v.y, s0.z += v.z, s0.w += v.w;
        s1.x += v.x * v.x, s1.y += v.y * v.y, s1.z += v.z * v.z, s1.w += v.w * v.w;
      }
    } else {
      const float4 mean = *reinterpret_cast<const float4*>(stats + c);
      const float4 rstd = *reinterpret_cast<const float4*>(stats + C + c);
      for (int64_t r = r0 + rl; r < r1; r += lanes) {
        const float4 yv = *reinterpret_cast<const float4*>(y + r * C + c);
        const float4 dv = *reinterpret_cast<const float4*>(dy + r * C + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
        const float g0 = yv.x > 0.f ? dv.x : 0.f, g1 = yv.y > 0.f ? dv.y : 0.f;
        const float g2 = yv.z > 0.f ? dv.z : 0.f, g3 = yv.w > 0.f ? dv.w : 0.f;
        s0.x += g0, s0.y += g1, s0.z += g2, s0.w += g3;
        s1.x += g0 * ((xv.x - mean.x) * rstd.x), s1.y += g1 * ((xv.y - mean.y) * rstd.y);
        s1.z += g2 * ((xv.z - mean.z) * rstd.z), s1.w += g3 * ((xv.w - mean.w) * rstd.w);
      }
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (rl != 0 || !live) return;
  for (int l = 1; l < lanes; ++l) {
    const float4 a = red[0][l * cw + lc], b = red[1][l * cw + lc];
    s0.x += a.x, s0.y += a.y, s0.z += a.z, s0.w += a.w;
    s1.x += b.x, s1.y += b.y, s1.z += b.z, s1.w += b.w;
  }
  *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * 2) * C + c) = s0;
  *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.y * 2 + 1) * C + c) = s1;
}

inline unsigned bn_channel_blocks(int C) {
  const int c4n = C / 4;
  return (unsigned)(c4n <= 256 ? 1 : (c4n + 255) / 256);
}

// sums the chunks in a fixed order (four interleaved runs per channel, combined 0..3).  mode 0: stats = (mean, rstd) with
// the biased batch variance, running statistics updated like nn.BatchNorm2d (momentum, unbiased variance).
// mode 1: out = (sum g, sum g xhat) = (dbeta, dgamma).  64 channels x 4 runs per workgroup.
__global__ __launch_bounds__(256) void bn_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                        float* __restrict__ run_mean, float* __restrict__ run_var,
                                                        int n_chunk, int64_t R, int C, float eps, float momentum,
                                                        int mode) {
  __shared__ float red[2][4][64];
  const int lc = threadIdx.x & 63, run = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lc;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
    for (int k = run; k < n_chunk; k += 4) {
      s0 += partial[((int64_t)k * 2) * C + c];
      s1 += partial[((int64_t)k * 2 + 1) * C + c];
    }
  red[0][run][lc] = s0;
  red[1][run][lc] = s1;
  __syncthreads();
  if (run != 0 || c >= C) return;
  s0 = (red[0][0][lc] + red[0][1][lc]) + (red[0][2][lc] + red[0][3][lc]);
  s1 = (red[1][0][lc] + red[1][1][lc]) + (red[1][2][lc] + red[1][3][lc]);
  if (mode == 0) {
    const double mean_d = (double)s0 / (double)R;
    const float mean = (float)mean_d;
    const float var = (float)fmax((double)s1 / (double)R - mean_d * mean_d, 0.0);
    out[c] = mean;
    out[C + c] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
      const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
    }
  } else {
    out[c] = s0;
    out[C + c] = s1;
  }
}

// forward: y = relu((x - mean) rstd gamma + beta);  backward: dx = gamma rstd (g - (S0 + xhat S1) / R), g = dy (y > 0)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ y_in,
                                                       const float* __restrict__ dy, const float* __restrict__ stats,
                                                       const float* __restrict__ sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out,
                                                       int64_t R, int C, int backward) {
  const int64_t n4 = R * C / 4;  // C % 4 == 0
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % C);
    const float4 xv = *reinterpret_cast<const float4*>(x + i * 4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
    if (!backward) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = fmaxf((xs[j] - stats[c + j]) * stats[C + c + j] * gamma[c + j] + beta[c + j], 0.f);
    } else {
      const float4 yv = *reinterpret_cast<const float4*>(y_in + i * 4);
      const float4 gv = *reinterpret_cast<const float4*>(dy + i * 4);
      const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
      const float inv = 1.0f / (float)R;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = ys[j] > 0.f ? gs[j] : 0.f;
        const float rstd = stats[C + c + j];
        const float xhat = (xs[j] - stats[c + j]) * rstd;
        o[j] = gamma[c + j] * rstd * (g - (sums[c + j] + xhat * sums[C + c + j]) * inv);
      }
    }
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------ im2col (k3 s2 p1)
__global__ __launch_bounds__(256) void im2col_s2_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H,
                                                        int W, int C, int ld) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, c4 = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * 9 * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4);
    int64_t r = i / c4;
    const int k = (int)(r % 9);
    r /= 9;
    const int wo = (int)(r % Wo);
    const int64_t bh = r / Wo;
    const int ho = (int)(bh % Ho), b = (int)(bh / Ho);
    const int h = 2 * ho + k / 3 - 1, w = 2 * wo + k % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h >= 0 && h < H && w >= 0 && w < W)
      v = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + h) * W + w) * C + c * 4);
    *reinterpret_cast<float4*>(col + r * ld + k * C + c * 4) = v;
  }
}

__global__ __launch_bounds__(256) void col2im_s2_kernel(const float* __restrict__ col, float* __restrict__ dx, int B,
                                                        int H, int W, int C, int ld) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, c4 = C / 4;
  const int64_t total = (int64_t)B * H * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4);
    int64_t r = i / c4;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H), b = (int)(r / H);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + 1 - kh;
      if (hh < 0 || (hh & 1) || hh / 2 >= Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w + 1 - kw;
        if (ww < 0 || (ww & 1) || ww / 2 >= Wo) continue;
        const float4 v = *reinterpret_cast<const float4*>(
            col + (((int64_t)b * Ho + hh / 2) * Wo + ww / 2) * ld + (kh * 3 + kw) * C + c * 4);
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = a;
  }
}

// ------------------------------------------------------------------------------------------------ GRU
// forward on gi = x W_ih^T + b_ih (B, T, 3H): keeps h_0..h_T (hs (B, T+1, H)) and per step [r | z | n | W_hn h + b_hn]
// (gates (B, T, 4H)).  One workgroup of 3H threads per item; w_hh_t (H, 3H) = W_hh transposed (coalesced matvec).
__global__ void gru_train_fwd_kernel(const float* __restrict__ gi_all, const float* __restrict__ w_hh_t,
                                     const float* __restrict__ b_hh, float* __restrict__ hs,
                                     float* __restrict__ gates, int T, int H) {
  extern __shared__ float sm_g[];  // h[H] | gh[3H]
  float* sh = sm_g;
  float* gh = sh + H;
  const int b = blockIdx.x, tid = threadIdx.x, G = 3 * H;
  for (int i = tid; i < H; i += blockDim.x) {
    sh[i] = 0.f;
    hs[(int64_t)b * (T + 1) * H + i] = 0.f;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (tid < G) {
      float c = 0.f;
      for (int i = 0; i < H; ++i) c = fmaf(w_hh_t[(int64_t)i * G + tid], sh[i], c);
      gh[tid] = c + b_hh[tid];
    }
    __syncthreads();
    if (tid < H) {
      const float* gi = gi_all + ((int64_t)b * T + t) * G;
      float* gt = gates + ((int64_t)b * T + t) * 4 * H;
      const float r = 1.0f / (1.0f + expf(-(gi[tid] + gh[tid])));
      const float z = 1.0f / (1.0f + expf(-(gi[H + tid] + gh[H + tid])));
      const float ghn = gh[2 * H + tid];
      const float n = tanhf(gi[2 * H + tid] + r * ghn);
      const float h = (1.0f - z) * n + z * sh[tid];
      gt[tid] = r, gt[H + tid] = z, gt[2 * H + tid] = n, gt[3 * H + tid] = ghn;
      sh[tid] = h;
      hs[((int64_t)b * (T + 1) + t + 1) * H + tid] = h;
    }
    __syncthreads();
  }
}

// BPTT from the gradient of the last hidden state: dgi (B, T, 3H) (gradient of the input projection) and dgh (B, T, 3H)
// (gradient of W_hh h + b_hh: dW_hh = dgh^T h_prev and db_hh = sum dgh are contractions over (b, t) done outside).
__global__ void gru_train_bwd_kernel(const float* __restrict__ dh_last, const float* __restrict__ w_hh,
                                     const float* __restrict__ hs, const float* __restrict__ gates,
                                     float* __restrict__ dgi, float* __restrict__ dgh, int T, int H) {
  extern __shared__ float sm_g[];  // dh[H] | dhz[H] | g[3H]
  float* dh = sm_g;
  float* dhz = dh + H;
  float* g = dhz + H;
  const int b = blockIdx.x, tid = threadIdx.x, G = 3 * H;
  for (int i = tid; i < H; i += blockDim.x) dh[i] = dh_last[(int64_t)b * H + i];
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    if (tid < H) {
      const float* gt = gates + ((int64_t)b * T + t) * 4 * H;
      const float r = gt[tid], z = gt[H + tid], n = gt[2 * H + tid], ghn = gt[3 * H + tid];
      const float hp = hs[((int64_t)b * (T + 1) + t) * H + tid];
      const float d = dh[tid];
      const float dn_pre = d * (1.0f - z) * (1.0f - n * n);
      const float dz_pre = d * (hp - n) * z * (1.0f - z);
      const float dr_pre = dn_pre * ghn * r * (1.0f - r);
      float* o = dgi + ((int64_t)b * T + t) * G;
      float* q = dgh + ((int64_t)b * T + t) * G;
      o[tid] = dr_pre, o[H + tid] = dz_pre, o[2 * H + tid] = dn_pre;
      const float dghn = dn_pre * r;
      q[tid] = dr_pre, q[H + tid] = dz_pre, q[2 * H + tid] = dghn;
      g[tid] = dr_pre, g[H + tid] = dz_pre, g[2 * H + tid] = dghn;
      dhz[tid] = d * z;
    }
    __syncthreads();
    if (tid < H) {  // dh_prev = dh z + W_hh^T dgh  (column tid of W_hh: lanes read consecutive addresses)
      float c = dhz[tid];
      for (int k = 0; k < G; ++k) c = fmaf(w_hh[(int64_t)k * H + tid], g[k], c);
      dh[tid] = c;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ style tokens
// one query per item against n_tok keys / values, n_head heads of dk = F / n_head:
// fwd: p[b, h, :] = softmax(q_h . k_h / sqrt(dk)), ctx[b, h*dk + d] = sum_t p v.  One workgroup per item.
__global__ __launch_bounds__(256) void token_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, float* __restrict__ p,
                                                             float* __restrict__ ctx, int n_tok, int F, int n_head) {
  extern __shared__ float sm_t[];  // q[F] | sc[n_head * n_tok]
  float* sq = sm_t;
  float* sc = sq + F;
  const int b = blockIdx.x, tid = threadIdx.x, dk = F / n_head;
  for (int i = tid; i < F; i += 256) sq[i] = q[(int64_t)b * F + i];
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)dk);
  for (int i = tid; i < n_head * n_tok; i += 256) {
    const int h = i / n_tok, t = i - h * n_tok;
    float a = 0.f;
    for (int d = 0; d < dk; ++d) a = fmaf(sq[h * dk + d], k[(int64_t)t * F + h * dk + d], a);
    sc[i] = a * scale;
  }
  __syncthreads();
  if (tid < n_head) {
    float m = -3.0e38f;
    for (int t = 0; t < n_tok; ++t) m = fmaxf(m, sc[tid * n_tok + t]);
    float s = 0.f;
    for (int t = 0; t < n_tok; ++t) {
      const float e = expf(sc[tid * n_tok + t] - m);
      sc[tid * n_tok + t] = e;
      s += e;
    }
    for (int t = 0; t < n_tok; ++t) sc[tid * n_tok + t] /= s;
  }
  __syncthreads();
  for (int i = tid; i < n_head * n_tok; i += 256) p[(int64_t)b * n_head * n_tok + i] = sc[i];
  for (int f = tid; f < F; f += 256) {
    const int h = f / dk;
    float a = 0.f;
    for (int t = 0; t < n_tok; ++t) a = fmaf(sc[h * n_tok + t], v[(int64_t)t * F + f], a);
    ctx[(int64_t)b * F + f] = a;
  }
}

// bwd: dq (B, F); dk_part / dv_part (B, n_tok, F) per item (summed over items outside, fixed order)
__global__ __launch_bounds__(256) void token_attn_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ q,
                                                             const float* __restrict__ k, const float* __restrict__ v,
                                                             const float* __restrict__ p, float* __restrict__ dq,
                                                             float* __restrict__ dk_part, float* __restrict__ dv_part,
                                                             int n_tok, int F, int n_head) {
  extern __shared__ float sm_t[];  // dctx[F] | q[F] | p[n_head * n_tok] | ds[n_head * n_tok]
  float* sd = sm_t;
  float* sq = sd + F;
  float* sp = sq + F;
  float* ds = sp + n_head * n_tok;
  const int b = blockIdx.x, tid = threadIdx.x, dk = F / n_head;
  for (int i = tid; i < F; i += 256) {
    sd[i] = dctx[(int64_t)b * F + i];
    sq[i] = q[(int64_t)b * F + i];
  }
  for (int i = tid; i < n_head * n_tok; i += 256) sp[i] = p[(int64_t)b * n_head * n_tok + i];
  __syncthreads();
  // dp[h, t] = dctx_h . v[t]_h
  for (int i = tid; i < n_head * n_tok; i += 256) {
    const int h = i / n_tok, t = i - h * n_tok;
    float a = 0.f;
    for (int d = 0; d < dk; ++d) a = fmaf(sd[h * dk + d], v[(int64_t)t * F + h * dk + d], a);
    ds[i] = a;
  }
  __syncthreads();
  if (tid < n_head) {  // softmax backward, scaled by 1 / sqrt(dk)
    float s = 0.f;
    for (int t = 0; t < n_tok; ++t) s += sp[tid * n_tok + t] * ds[tid * n_tok + t];
    const float scale = 1.0f / sqrtf((float)dk);
    for (int t = 0; t < n_tok; ++t) ds[tid * n_tok + t] = scale * sp[tid * n_tok + t] * (ds[tid * n_tok + t] - s);
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    const int h = f / dk;
    float a = 0.f;
    for (int t = 0; t < n_tok; ++t) a = fmaf(ds[h * n_tok + t], k[(int64_t)t * F + f], a);
    dq[(int64_t)b * F + f] = a;
  }
  for (int i = tid; i < n_tok * F; i += 256) {
    const int t = i / F, f = i - t * F, h = f / dk;
    dk_part[(int64_t)b * n_tok * F + i] = ds[h * n_tok + t] * sq[f];
    dv_part[(int64_t)b * n_tok * F + i] = sp[h * n_tok + t] * sd[f];
  }
}

}  // namespace

extern "C" int srn_bn_chunks(int64_t rows) {
  const int64_t per = bn_rows_per_chunk(rows);
  return (int)((rows + per - 1) / per);
}


extern "C" int srn_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                               float* partial, float* stats, float* y, int64_t rows, int C, float eps, float momentum,
                               void* stream) {
  SRN_CHECK_ARG(x && gamma && beta && partial && stats && y && rows > 0 && C > 0 && C % 4 == 0, "bn_relu_fwd: bad args");
  SRN_CHECK_ARG((run_mean == nullptr) == (run_var == nullptr), "bn_relu_fwd: running statistics come together");
  const int chunks = srn_bn_chunks(rows);
  SRN_CHECK_ARG(chunks < 65536, "bn_relu_fwd: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const unsigned cb = bn_channel_blocks(C);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(cb, chunks), dim3(256), 0, st, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, partial, rows, C, 0);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, (const float*)partial, stats,
                     run_mean, run_var, chunks, rows, C, eps, momentum, 0);
  const int64_t n4 = rows * C / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                     (const float*)stats, (const float*)nullptr, gamma, beta, y, rows, C, 0);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* stats, const float* gamma,
                               float* partial, float* sums, float* dx, int64_t rows, int C, void* stream) {
  SRN_CHECK_ARG(x && y && dy && stats && gamma && partial && sums && dx && rows > 0 && C > 0 && C % 4 == 0,
                "bn_relu_bwd: bad args");
  const int chunks = srn_bn_chunks(rows);
  SRN_CHECK_ARG(chunks < 65536, "bn_relu_bwd: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const unsigned cb = bn_channel_blocks(C);
  hipLaunchKernelGGL(bn_partial_kernel, dim3(cb, chunks), dim3(256), 0, st, x, y, dy, stats, partial, rows, C, 1);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, (const float*)partial, sums,
                     (float*)nullptr, (float*)nullptr, chunks, rows, C, 0.f, 0.f, 1);
  const int64_t n4 = rows * C / 4;
  const unsigned blocks = (unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, x, y, dy, stats, (const float*)sums, gamma,
                     (const float*)nullptr, dx, rows, C, 1);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_im2col_s2(const float* x, float* col, int B, int H, int W, int C, int ld, void* stream) {
  SRN_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld >= 9 * C && ld % 4 == 0, "im2col_s2: bad args");
  const int64_t total = (int64_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * 9 * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(im2col_s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, col, B, H, W, C, ld);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_col2im_s2(const float* col, float* dx, int B, int H, int W, int C, int ld, void* stream) {
  SRN_CHECK_ARG(col && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld >= 9 * C && ld % 4 == 0, "col2im_s2: bad args");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(col2im_s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, col, dx, B, H, W, C, ld);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gru_train_fwd(const float* gi, const float* w_hh_t, const float* b_hh, float* hs, float* gates, int B,
                                 int T, int H, void* stream) {
  SRN_CHECK_ARG(gi && w_hh_t && b_hh && hs && gates && B > 0 && T > 0 && H > 0 && 3 * H <= 1024, "gru_train_fwd: bad args");
  const int threads = ((3 * H + 63) / 64) * 64;
  hipLaunchKernelGGL(gru_train_fwd_kernel, dim3(B), dim3(threads), (size_t)(4 * H) * sizeof(float), (hipStream_t)stream,
                     gi, w_hh_t, b_hh, hs, gates, T, H);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gru_train_bwd(const float* dh_last, const float* w_hh, const float* hs, const float* gates,
                                 float* dgi, float* dgh, int B, int T, int H, void* stream) {
  SRN_CHECK_ARG(dh_last && w_hh && hs && gates && dgi && dgh && B > 0 && T > 0 && H > 0 && 3 * H <= 1024,
                "gru_train_bwd: bad args");
  const int threads = ((H + 63) / 64) * 64;
  hipLaunchKernelGGL(gru_train_bwd_kernel, dim3(B), dim3(threads), (size_t)(5 * H) * sizeof(float), (hipStream_t)stream,
                     dh_last, w_hh, hs, gates, dgi, dgh, T, H);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_token_attn_fwd(const float* q, const float* k, const float* v, float* p, float* ctx, int B, int n_tok,
                                  int F, int n_head, void* stream) {
  SRN_CHECK_ARG(q && k && v && p && ctx && B > 0 && n_tok > 0 && F > 0 && n_head > 0 && n_head <= 256 &&
                    F % n_head == 0, "token_attn_fwd: bad args");
  const size_t smem = (size_t)(F + n_head * n_tok) * sizeof(float);
  SRN_CHECK_ARG(smem <= 64 * 1024, "token_attn_fwd: too large for LDS");
  hipLaunchKernelGGL(token_attn_fwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, q, k, v, p, ctx, n_tok, F,
                     n_head);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_token_attn_bwd(const float* dctx, const float* q, const float* k, const float* v, const float* p,
                                  float* dq, float* dk_part, float* dv_part, int B, int n_tok, int F, int n_head,
                                  void* stream) {
  SRN_CHECK_ARG(dctx && q && k && v && p && dq && dk_part && dv_part && B > 0 && n_tok > 0 && F > 0 && n_head > 0 &&
                    n_head <= 256 && F % n_head == 0, "token_attn_bwd: bad args");
  const size_t smem = (size_t)(2 * F + 2 * n_head * n_tok) * sizeof(float);
  SRN_CHECK_ARG(smem <= 64 * 1024, "token_attn_bwd: too large for LDS");
  hipLaunchKernelGGL(token_attn_bwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, dctx, q, k, v, p, dq, dk_part,
                     dv_part, n_tok, F, n_head);
  SRN_CHECK_LAUNCH();
  return 0;
}
