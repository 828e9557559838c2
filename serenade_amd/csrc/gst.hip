// gst.hip — GST style encoder stages (run once per utterance, ~0.3 % of the path's FLOPs):
// Conv2d(k3,s2,p1)+BatchNorm2d(eval)+ReLU on NHWC tensors, the GRU's last hidden state and the
// style-token multi-head attention.  Reference: serenade/modules/gst/style_encoder.py:142-191,235-252 and
// serenade/modules/gst/attention.py:110-184,298-300.
#include "common.h"

namespace {

// One workgroup per (b, ho): the 3 input rows it needs are staged zero-padded in LDS
// ([3][W + 2][Ci]); each thread owns output channels co = tid, tid + 256, ... and keeps the whole
// output row (WO_MAX accumulators) in registers, so every weight is read exactly once per workgroup.
template <int WO_MAX>
__global__ __launch_bounds__(256) void conv2d_bn_relu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bn_scale,
                                                             const float* __restrict__ bn_shift,
                                                             float* __restrict__ y, int H, int W, int Ci, int Co,
                                                             int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float srow[];  // [3][(W + 2)][Ci]
  const int b = blockIdx.y, ho = blockIdx.x;
  const int Wp = W + 2;
  const int row_elems = Wp * Ci;
  for (int idx = threadIdx.x; idx < 3 * row_elems; idx += 256) {
    const int kh = idx / row_elems;
    const int rem = idx - kh * row_elems;
    const int wp = rem / Ci, ci = rem - wp * Ci;
    const int h = 2 * ho + kh - 1, wi = wp - 1;
    float v = 0.f;
    if (h >= 0 && h < H && wi >= 0 && wi < W) v = x[(((int64_t)b * H + h) * W + wi) * Ci + ci];
    srow[idx] = v;
  }
  __syncthreads();
  for (int co = threadIdx.x; co < Co; co += 256) {
    float acc[WO_MAX];
#pragma unroll
    for (int i = 0; i < WO_MAX; ++i) acc[i] = 0.f;
    const float* wc = w + (int64_t)co * 9 * Ci;
    for (int kh = 0; kh < 3; ++kh) {
      for (int kw = 0; kw < 3; ++kw) {
        const float* wk = wc + (kh * 3 + kw) * Ci;
        const float* xr = srow + kh * row_elems + kw * Ci;  // input column 2*wo + kw (padded index)
        for (int ci = 0; ci < Ci; ++ci) {
          const float wv = wk[ci];
#pragma unroll
          for (int wo = 0; wo < WO_MAX; ++wo)
            if (wo < Wo) acc[wo] = fmaf(xr[(2 * wo) * Ci + ci], wv, acc[wo]);
        }
      }
    }
    const float sc = bn_scale[co], sh = bn_shift[co];
#pragma unroll
    for (int wo = 0; wo < WO_MAX; ++wo)
      if (wo < Wo) y[(((int64_t)b * Ho + ho) * Wo + wo) * Co + co] = fmaxf(acc[wo] * sc + sh, 0.f);
  }
}

// GRU, batch_first, one layer, last hidden state.  One workgroup per batch item, 3H threads
// (one per gate row).  Gates r, z, n (torch order).
__global__ void gru_last_kernel(const float* __restrict__ xs, const float* __restrict__ w_ih,
                                const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                const float* __restrict__ b_hh, float* __restrict__ hout, int T, int I, int H) {
  extern __shared__ float sm[];  // x[I] | h[H] | gi[3H] | gh[3H]
  float* sx = sm;
  float* sh = sx + I;
  float* gi = sh + H;
  float* gh = gi + 3 * H;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < H; i += blockDim.x) sh[i] = 0.f;
  for (int t = 0; t < T; ++t) {
    for (int i = tid; i < I; i += blockDim.x) sx[i] = xs[((int64_t)b * T + t) * I + i];
    __syncthreads();
    if (tid < 3 * H) {
      const float* wi = w_ih + (int64_t)tid * I;
      float a = 0.f;
      for (int i = 0; i < I; ++i) a = fmaf(wi[i], sx[i], a);
      gi[tid] = a + b_ih[tid];
      const float* wh = w_hh + (int64_t)tid * H;
      float c = 0.f;
      for (int i = 0; i < H; ++i) c = fmaf(wh[i], sh[i], c);
      gh[tid] = c + b_hh[tid];
    }
    __syncthreads();
    if (tid < H) {
      const float r = 1.0f / (1.0f + expf(-(gi[tid] + gh[tid])));
      const float z = 1.0f / (1.0f + expf(-(gi[H + tid] + gh[H + tid])));
      const float n = tanhf(gi[2 * H + tid] + r * gh[2 * H + tid]);
      sh[tid] = (1.0f - z) * n + z * sh[tid];
    }
    __syncthreads();
  }
  for (int i = tid; i < H; i += blockDim.x) hout[(int64_t)b * H + i] = sh[i];
}

// Style-token attention: one workgroup (256 threads) per batch item.  F <= 256, n_tok <= 64.
__global__ __launch_bounds__(256) void style_token_attention_kernel(
    const float* __restrict__ ref, const float* __restrict__ embs, const float* __restrict__ wq,
    const float* __restrict__ bq, const float* __restrict__ wk, const float* __restrict__ bk,
    const float* __restrict__ wv, const float* __restrict__ bv, const float* __restrict__ wo,
    const float* __restrict__ bo, float* __restrict__ out, int Dq, int n_tok, int dk_in, int F, int n_head) {
  extern __shared__ float sm[];
  float* q = sm;                   // [F]
  float* toks = q + F;             // [n_tok][dk_in] tanh'ed
  float* kk = toks + n_tok * dk_in;  // [n_tok][F]
  float* vv = kk + n_tok * F;      // [n_tok][F]
  float* sc = vv + n_tok * F;      // [n_head][n_tok]
  float* ctx = sc + n_head * n_tok;  // [F]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int dk = F / n_head;
  for (int i = tid; i < n_tok * dk_in; i += 256) toks[i] = tanhf(embs[i]);
  for (int f = tid; f < F; f += 256) {
    float a = 0.f;
    for (int i = 0; i < Dq; ++i) a = fmaf(ref[(int64_t)b * Dq + i], wq[(int64_t)f * Dq + i], a);
    q[f] = a + bq[f];
  }
  __syncthreads();
  for (int idx = tid; idx < n_tok * F; idx += 256) {
    const int t = idx / F, f = idx - t * F;
    float a = 0.f, c = 0.f;
    for (int i = 0; i < dk_in; ++i) {
      a = fmaf(toks[t * dk_in + i], wk[f * dk_in + i], a);
      c = fmaf(toks[t * dk_in + i], wv[f * dk_in + i], c);
    }
    kk[idx] = a + bk[f];
    vv[idx] = c + bv[f];
  }
  __syncthreads();
  const float inv = 1.0f / sqrtf((float)dk);
  for (int idx = tid; idx < n_head * n_tok; idx += 256) {
    const int h = idx / n_tok, t = idx - h * n_tok;
    float a = 0.f;
    for (int d = 0; d < dk; ++d) a = fmaf(q[h * dk + d], kk[t * F + h * dk + d], a);
    sc[idx] = a * inv;
  }
  __syncthreads();
  if (tid < n_head) {
    float mx = -INFINITY;
    for (int t = 0; t < n_tok; ++t) mx = fmaxf(mx, sc[tid * n_tok + t]);
    float s = 0.f;
    for (int t = 0; t < n_tok; ++t) {
      const float e = expf(sc[tid * n_tok + t] - mx);
      sc[tid * n_tok + t] = e;
      s += e;
    }
    for (int t = 0; t < n_tok; ++t) sc[tid * n_tok + t] /= s;
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    const int h = f / dk;
    float a = 0.f;
    for (int t = 0; t < n_tok; ++t) a = fmaf(sc[h * n_tok + t], vv[t * F + f], a);
    ctx[f] = a;
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    float a = 0.f;
    for (int i = 0; i < F; ++i) a = fmaf(ctx[i], wo[(int64_t)f * F + i], a);
    out[(int64_t)b * F + f] = a + bo[f];
  }
}

// GRU recurrence on a PRECOMPUTED input projection gi = x W_ih^T + b_ih (one srn_conv_gemm over all (b, t) rows,
// spread over the chip) -- the part that is inherently sequential is only h -> W_hh h, done here by one workgroup per
// batch item with W_hh TRANSPOSED (w_hh_t [H][3H]) so that gate row tid reads consecutive addresses across lanes.
// Round 1's gru_last_kernel walked both matrices with a 4 KB lane stride from one workgroup (220 us for 4 steps).
__global__ void gru_recur_last_kernel(const float* __restrict__ gi_all, const float* __restrict__ w_hh_t,
                                      const float* __restrict__ b_hh, float* __restrict__ hout, int T, int H) {
  extern __shared__ float sm[];  // h[H] | gh[3H]
  float* sh = sm;
  float* gh = sh + H;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int G = 3 * H;
  for (int i = tid; i < H; i += blockDim.x) sh[i] = 0.f;
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
      const float r = 1.0f / (1.0f + expf(-(gi[tid] + gh[tid])));
      const float z = 1.0f / (1.0f + expf(-(gi[H + tid] + gh[H + tid])));
      const float n = tanhf(gi[2 * H + tid] + r * gh[2 * H + tid]);
      sh[tid] = (1.0f - z) * n + z * sh[tid];
    }
    __syncthreads();
  }
  for (int i = tid; i < H; i += blockDim.x) hout[(int64_t)b * H + i] = sh[i];
}

// Style-token attention on PRECOMPUTED keys / values: K = tanh(embs) W_k^T + b_k and V likewise do not depend on the
// input, so they are formed once at weight-packing time ((n_tok, F) each); the query / output projections read
// TRANSPOSED weights (wq_t [Dq][F], wo_t [F][F]) so lanes read consecutive addresses.  One workgroup per batch item.
__global__ __launch_bounds__(256) void style_token_attention_kv_kernel(
    const float* __restrict__ ref, const float* __restrict__ wq_t, const float* __restrict__ bq,
    const float* __restrict__ kk, const float* __restrict__ vv, const float* __restrict__ wo_t,
    const float* __restrict__ bo, float* __restrict__ out, int Dq, int n_tok, int F, int n_head) {
  extern __shared__ float sm[];
  float* q = sm;                       // [F]
  float* sc = q + F;                   // [n_head][n_tok]
  float* ctx = sc + n_head * n_tok;    // [F]
  float* sref = ctx + F;               // [Dq]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int dk = F / n_head;
  for (int i = tid; i < Dq; i += 256) sref[i] = ref[(int64_t)b * Dq + i];
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    float a = 0.f;
    for (int i = 0; i < Dq; ++i) a = fmaf(sref[i], wq_t[(int64_t)i * F + f], a);
    q[f] = a + bq[f];
  }
  __syncthreads();
  const float inv = 1.0f / sqrtf((float)dk);
  for (int idx = tid; idx < n_head * n_tok; idx += 256) {
    const int h = idx / n_tok, t = idx - h * n_tok;
    float a = 0.f;
    for (int d = 0; d < dk; ++d) a = fmaf(q[h * dk + d], kk[t * F + h * dk + d], a);
    sc[idx] = a * inv;
  }
  __syncthreads();
  if (tid < n_head) {
    float mx = -INFINITY;
    for (int t = 0; t < n_tok; ++t) mx = fmaxf(mx, sc[tid * n_tok + t]);
    float s = 0.f;
    for (int t = 0; t < n_tok; ++t) {
      const float e = expf(sc[tid * n_tok + t] - mx);
      sc[tid * n_tok + t] = e;
      s += e;
    }
    for (int t = 0; t < n_tok; ++t) sc[tid * n_tok + t] /= s;
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    const int h = f / dk;
    float a = 0.f;
    for (int t = 0; t < n_tok; ++t) a = fmaf(sc[h * n_tok + t], vv[t * F + f], a);
    ctx[f] = a;
  }
  __syncthreads();
  for (int f = tid; f < F; f += 256) {
    float a = 0.f;
    for (int i = 0; i < F; ++i) a = fmaf(ctx[i], wo_t[(int64_t)i * F + f], a);
    out[(int64_t)b * F + f] = a + bo[f];
  }
}

}  // namespace

extern "C" int srn_gru_recur_last(const float* gi, const float* w_hh_t, const float* b_hh, float* h, int B, int T,
                                  int H, void* stream) {
  SRN_CHECK_ARG(gi && w_hh_t && b_hh && h && B > 0 && T > 0 && H > 0 && 3 * H <= 1024, "gru_recur_last: bad args");
  const int threads = ((3 * H + 63) / 64) * 64;
  hipLaunchKernelGGL(gru_recur_last_kernel, dim3(B), dim3(threads), (size_t)(4 * H) * sizeof(float),
                     (hipStream_t)stream, gi, w_hh_t, b_hh, h, T, H);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_style_token_attention_kv(const float* ref, const float* wq_t, const float* bq, const float* k,
                                            const float* v, const float* wo_t, const float* bo, float* out, int B,
                                            int Dq, int n_tok, int F, int n_head, void* stream) {
  SRN_CHECK_ARG(ref && wq_t && bq && k && v && wo_t && bo && out, "style_token_attention_kv: null");
  SRN_CHECK_ARG(B > 0 && F > 0 && n_head > 0 && F % n_head == 0 && n_tok > 0 && Dq > 0,
                "style_token_attention_kv: bad sizes");
  const size_t smem = (size_t)(2 * F + n_head * n_tok + Dq) * sizeof(float);
  SRN_CHECK_ARG(smem <= 64 * 1024, "style_token_attention_kv: too large for LDS");
  hipLaunchKernelGGL(style_token_attention_kv_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, ref, wq_t, bq, k,
                     v, wo_t, bo, out, Dq, n_tok, F, n_head);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_conv2d_bn_relu(const float* x, const float* w, const float* bn_scale, const float* bn_shift,
                                  float* y, int B, int H, int W, int Ci, int Co, void* stream) {
  SRN_CHECK_ARG(x && w && bn_scale && bn_shift && y && B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0,
                "conv2d_bn_relu: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t smem = (size_t)3 * (W + 2) * Ci * sizeof(float);
  SRN_CHECK_ARG(smem <= 160 * 1024, "conv2d_bn_relu: input rows (%zu B) exceed LDS", smem);
  dim3 grid(Ho, B);
  hipStream_t st = (hipStream_t)stream;
#define SRN_C2D(WOM)                                                                                              \
  do {                                                                                                            \
    static SrnSmemAttr smem_attr;                                                                                 \
    if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv2d_bn_relu_kernel<WOM>), 160 * 1024))   \
      return e;                                                                                                   \
    hipLaunchKernelGGL(conv2d_bn_relu_kernel<WOM>, grid, dim3(256), smem, st, x, w, bn_scale, bn_shift, y, H, W, Ci, \
                       Co, Ho, Wo);                                                                               \
  } while (0)
  if (Wo <= 2) SRN_C2D(2);
  else if (Wo <= 5) SRN_C2D(5);
  else if (Wo <= 10) SRN_C2D(10);
  else if (Wo <= 20) SRN_C2D(20);
  else if (Wo <= 40) SRN_C2D(40);
  else {
    srn_set_error("conv2d_bn_relu: Wo=%d > 40 unsupported", Wo);
    return -1;
  }
#undef SRN_C2D
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_gru_last(const float* xs, const float* w_ih, const float* w_hh, const float* b_ih,
                            const float* b_hh, float* h, int B, int T, int I, int H, void* stream) {
  SRN_CHECK_ARG(xs && w_ih && w_hh && b_ih && b_hh && h && B > 0 && T > 0 && I > 0 && H > 0 && 3 * H <= 1024,
                "gru_last: bad args");
  const int threads = ((3 * H + 63) / 64) * 64;
  const size_t smem = (size_t)(I + H + 6 * H) * sizeof(float);
  hipLaunchKernelGGL(gru_last_kernel, dim3(B), dim3(threads), smem, (hipStream_t)stream, xs, w_ih, w_hh, b_ih, b_hh, h,
                     T, I, H);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_style_token_attention(const float* ref, const float* embs, const float* wq, const float* bq,
                                         const float* wk, const float* bk, const float* wv, const float* bv,
                                         const float* wo, const float* bo, float* out, int B, int Dq, int n_tok,
                                         int dk_in, int F, int n_head, void* stream) {
  SRN_CHECK_ARG(ref && embs && wq && bq && wk && bk && wv && bv && wo && bo && out, "style_token_attention: null");
  SRN_CHECK_ARG(B > 0 && F > 0 && n_head > 0 && F % n_head == 0 && n_tok > 0, "style_token_attention: bad sizes");
  const size_t smem = (size_t)(F + n_tok * dk_in + 2 * n_tok * F + n_head * n_tok + F) * sizeof(float);
  SRN_CHECK_ARG(smem <= 160 * 1024, "style_token_attention: too large for LDS");
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&style_token_attention_kernel), 160 * 1024)) return e;
  hipLaunchKernelGGL(style_token_attention_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, ref, embs, wq, bq, wk,
                     bk, wv, bv, wo, bo, out, Dq, n_tok, dk_in, F, n_head);
  SRN_CHECK_LAUNCH();
  return 0;
}
