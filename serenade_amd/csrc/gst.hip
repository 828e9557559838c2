// gst.hip — the tail of the GST style encoder (run once per utterance): the GRU recurrence on a precomputed input
// projection and the style-token multi-head attention on precomputed keys / values.  (The Conv2d + BatchNorm + ReLU
// layers are srn_conv_gemm launches, one per kernel row: models.StyleEncoder.build_ops.)  Reference: serenade/modules/gst/style_encoder.py:142-191,235-252 and
// serenade/modules/gst/attention.py:110-184,298-300.
#include "common.h"

namespace {

// GRU recurrence on a PRECOMPUTED input projection gi = x W_ih^T + b_ih (one srn_conv_gemm over all (b, t) rows,
// spread over the chip) -- the part that is inherently sequential is only h -> W_hh h, done here by one workgroup per
// batch item with W_hh TRANSPOSED (w_hh_t [H][3H]) so that gate row tid reads consecutive addresses across lanes.
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
