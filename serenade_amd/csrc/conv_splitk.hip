// conv_splitk.hip -- split-K for contractions whose tile grid cannot fill the chip.
//
// At B = 1 / short utterances the N = 512 convs and projections of the UNet (K = 512 ... 3072) are 32-64 tiles of
// 64 x 64: a quarter of the CUs walk 16-96 dependent k-steps each (measured 31 us average over 413 launches per
// B = 1 x T = 256 Euler step, 55 % of the GPU time).  Here the (tap, channel) steps are sliced over `ksplit` extra
// workgroup sets (conv_fast.hip, ksplit > 1) that store raw fp32 partial sums to a workspace slab
// [slice][z][T_out][N]; this file's kernel sums the slices IN SLICE ORDER (bit-reproducible, no atomics) and applies
// the one epilogue every contraction has (alpha, bias, output mask, residual add / axpy, second residual, post op,
// strided rows, GroupNorm partial sums).  GEGLU and the transposed tail are never split (their GEMMs are wide).
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int MAX_KSPLIT = 8;

// one workgroup per (z, 32-row, 32-column) block: thread -> (row tid / 8, 4 columns at 4 (tid % 8)); the block is
// exactly one GroupNorm partial-sum tile
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SrnConvParams p, const int ksplit, const int m32,
                                                            const int n32) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int nt = bid % n32;
  bid /= n32;
  const int mt = bid % m32;
  const int z = bid / m32;
  const int zb = z / p.n_head;
  const int zh = z - zb * p.n_head;
  const int row = mt * 32 + (tid >> 3);
  const int col = nt * 32 + (tid & 7) * 4;
  const bool ok = row < p.T_out && col < p.N;  // N % 4 == 0 (srn_splitk_plan)
  const int64_t Z = (int64_t)p.n_batch * p.n_head;
  const int64_t slab = Z * p.T_out * p.N;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    // all (up to MAX_KSPLIT) slices' loads are issued back to back from clamped addresses and summed in slice order:
    // a run-time-length load loop pays one dependent memory round trip per slice (measured 5.9 us per call at B = 1)
    const float* w = reinterpret_cast<const float*>(p.ws) + ((int64_t)z * p.T_out + row) * p.N + col;
    float4 q[MAX_KSPLIT];
#pragma unroll
    for (int s = 0; s < MAX_KSPLIT; ++s) q[s] = *reinterpret_cast<const float4*>(w + min(s, ksplit - 1) * slab);
#pragma unroll
    for (int s = 0; s < MAX_KSPLIT; ++s) {
      const float on = s < ksplit ? 1.f : 0.f;
      v[0] += on * q[s].x, v[1] += on * q[s].y, v[2] += on * q[s].z, v[3] += on * q[s].w;
    }
  }
  int len_out = p.T_out;
  if (p.len_out) len_out = min(p.len_out[zb], p.T_out);
  const int64_t orow = (int64_t)row * p.out_t_stride + p.out_t_off;
  float s1 = 0.f, s2 = 0.f;
  if (ok) {
    float* out = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs + orow * p.ld_out + col;
    const float* res = p.res ? p.res + (int64_t)zb * p.res_bs + (int64_t)zh * p.res_hs + orow * p.ld_res + col : nullptr;
    const float* res2 = p.res2 ? p.res2 + (int64_t)zb * p.res2_bs + orow * p.ld_res2 + col : nullptr;
    float rv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // residuals first: they may alias out
      if (res) rv[j] = res[j];
      if (res2) qv[j] = res2[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = v[j] * p.alpha + (p.bias ? p.bias[col + j] : 0.f);
      if (row >= len_out) x = 0.f;
      if (p.res_mode == SRN_RES_ADD) x += rv[j];
      else if (p.res_mode == SRN_RES_AXPY) x = rv[j] + p.beta * x;
      if (res2) x += qv[j];
      if (p.post == SRN_POST_DIV) x = x / p.post_div;
      else if (p.post == SRN_POST_TANH) x = tanhf(x);
      else if (p.post == SRN_POST_RELU) x = fmaxf(x, 0.f);
      else if (p.post == SRN_POST_LEAKY) x = x > 0.f ? x : x * p.post_div;
      out[j] = x;
      s1 += x;
      s2 += x * x;
    }
  }
  if (p.gn_partials) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) {
      red[(tid >> 6) * 2] = s1;
      red[(tid >> 6) * 2 + 1] = s2;
    }
    __syncthreads();
    const int gn_mt = (p.T_out + 31) / 32, gn_nt = p.N / 32;
    if (tid == 0 && mt < gn_mt && nt < gn_nt) {
      float* gp = p.gn_partials + (((int64_t)zb * gn_mt + mt) * gn_nt + nt) * 2;
      gp[0] = (red[0] + red[2]) + (red[4] + red[6]);
      gp[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
  }
}

}  // namespace

// K slices for this launch (1 = leave it alone).  Only shapes conv_fast.hip takes, with a plain or residual epilogue.
int srn_splitk_plan(const SrnConvParams& p) {
  if (p.geglu || p.out_tr != nullptr || p.w_nmajor || p.N % 4 != 0) return 1;
  if (p.C_in % 32 != 0 || p.C_in0 % 32 != 0) return 1;
  const bool wpl = ((p.precision == SRN_PREC_BF16X3 && p.w_hi != nullptr) ||
                    (p.precision == SRN_PREC_BF16X6 && p.w_hi != nullptr && p.w_lo != nullptr)) &&
                   p.w_bs == 0 && p.w_hs == 0;
  if (!wpl && (p.C_w != p.C_in || p.ldw < p.n_taps * p.C_in)) return 1;
  if (p.gn_partials && p.n_head != 1) return 1;
  const int steps = p.n_taps * (p.C_in / 32);
  // measured at B = 1 (1 workgroup per CU): ~4 us launch ramp + ~0.5 us per dependent k-step unsplit, vs ramp +
  // steps / ks + a ~3 us reduce launch when split -- below ~24 steps the reduce eats the gain
  if (steps < 24) return 1;
  const int64_t tiles = (int64_t)p.n_batch * p.n_head * ((p.T_out + 63) / 64) * ((p.N + 63) / 64);
  if (tiles > 192) return 1;
  // one workgroup per CU: conv_f32.hip's two-steps-ahead tile covers its own latency, and a second workgroup on a CU
  // halves both's matrix rate (B = 1 x T = 256 in place, same box: 15.44-15.52 ms at 256, 15.67 at round 3's 448 -- the
  // double-buffered conv_fast.hip tile wanted ~1.75 per CU --, 15.49 at 288, 16.0-16.3 at 192 / 224 / 320)
  int ks = (int)(256 / tiles);
  ks = ks > MAX_KSPLIT ? MAX_KSPLIT : ks;
  ks = ks > steps / 4 ? steps / 4 : ks;  // at least four steps per slice
  if (ks < 2) return 1;
  const int per = (steps + ks - 1) / ks;
  return (steps + per - 1) / per;  // no empty slice
}

int64_t srn_splitk_bytes(const SrnConvParams& p, int ksplit) {
  return (int64_t)ksplit * p.n_batch * p.n_head * p.T_out * p.N * (int64_t)sizeof(float);
}

int srn_splitk_reduce(const SrnConvParams& p, int ksplit, hipStream_t stream) {
  const int m32 = (p.T_out + 31) / 32, n32 = (p.N + 31) / 32;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m32 * n32;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "splitk_reduce: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, ksplit, m32, n32);
  SRN_CHECK_LAUNCH();
  return 0;
}
