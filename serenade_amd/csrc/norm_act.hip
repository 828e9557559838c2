// norm_act.hip — the HBM-bound stages between the contractions (gfx950): GroupNorm-apply + Mish + mask,
// the ResnetBlock1D tail (GroupNorm + Mish + residual + speaker-conditional LayerNorm), LayerNorm, masked
// row softmax, and small layout / elementwise helpers.  All tensors are channels-last fp32; every kernel
// moves 16 B per lane (float4) along the contiguous channel axis and reduces with wavefront shuffles.
#include "common.h"
#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics from the per-(32 rows x 32 cols) partials the conv epilogue wrote.
// partials: [b][gn_mt][gn_nt][2];  out: mean[g], rstd[g] in LDS.  Called by all 256 threads.
// n_rows: rows the statistics run over -- T (the reference's batched semantics: padded frames included), or the item's
// own length when the producing conv zeroed its padded rows (exact ragged batches: the B = 1 result of every item).
__device__ __forceinline__ void group_stats(const float* __restrict__ partials, int b, int T, int C, int groups,
                                            float eps, float* s_mean, float* s_rstd, int n_rows) {
  const int gn_mt = (T + 31) / 32;
  const int gn_nt = C / 32;
  const int nt_per_g = (C / groups) / 32;
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
      const double cnt = (double)max(n_rows, 1) * (double)(C / groups);
      const double mean = s1 / cnt;
      double var = s2 / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
}

constexpr int GN_ROWS = 8;  // rows per workgroup (small: every workgroup first re-reduces the group statistics, more of them overlap that prefix)
constexpr int SMALL_GRID = 512;  // workgroups below which the row kernels take half as many rows each (B = 1: latency, not bandwidth)
constexpr int GN_PRE = 4;   // float4 per thread fetched ahead of the statistics (GN_ROWS rows of 512 channels)

__global__ __launch_bounds__(256) void gn_mish_apply_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ partials,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ time_bias,
                                                            const int32_t* __restrict__ lens, float* __restrict__ y,
                                                            int T, int C, int groups, float eps, int64_t tb_bs,
                                                            int valid_stats, int rows_per_wg) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_wg;
  if (time_bias) time_bias += (int64_t)b * tb_bs;
  const int len = lens ? min(lens[b], T) : T;
  const int c4n = C / 4;
  const int cpg = C / groups;
  const int rows = min(rows_per_wg, T - r0);
  const int total = rows * c4n;
  const int64_t base = ((int64_t)b * T + r0) * C;
  // the rows' values are on their way while the statistics are reduced (one memory round trip instead of two in a row)
  float4 pre[GN_PRE];
#pragma unroll
  for (int k = 0; k < GN_PRE; ++k) {
    const int idx = threadIdx.x + 256 * k;
    pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
      const int r = idx / c4n;
      if (r0 + r < len) pre[k] = *reinterpret_cast<const float4*>(x + base + (int64_t)r * C + (idx - r * c4n) * 4);
    }
  }
  group_stats(partials, b, T, C, groups, eps, s_mean, s_rstd, valid_stats ? len : T);
  auto one = [&](const int idx, const float4 v) {
    const int r = idx / c4n;
    const int c = (idx - r * c4n) * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < len) {
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
      const float4 be = *reinterpret_cast<const float4*>(beta + c);
      const int g = c / cpg;
      const float m = s_mean[g], rs = s_rstd[g];
      o.x = srn_mish((v.x - m) * rs * ga.x + be.x);
      o.y = srn_mish((v.y - m) * rs * ga.y + be.y);
      o.z = srn_mish((v.z - m) * rs * ga.z + be.z);
      o.w = srn_mish((v.w - m) * rs * ga.w + be.w);
      if (time_bias) {
        const float4 tb = *reinterpret_cast<const float4*>(time_bias + c);
        o.x += tb.x;
        o.y += tb.y;
        o.z += tb.z;
        o.w += tb.w;
      }
    }
    *reinterpret_cast<float4*>(y + base + (int64_t)r * C + c) = o;
  };
#pragma unroll
  for (int k = 0; k < GN_PRE; ++k)
    if ((int)threadIdx.x + 256 * k < total) one(threadIdx.x + 256 * k, pre[k]);
  for (int idx = threadIdx.x + 256 * GN_PRE; idx < total; idx += 256) {
    const int r = idx / c4n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < len) v = *reinterpret_cast<const float4*>(x + base + (int64_t)r * C + (idx - r * c4n) * 4);
    one(idx, v);
  }
}

// ------------------------------------------------------------------------------------------------
// One wavefront per frame; C <= 1024 and C % 4 == 0 (up to MAXV float4 per lane).
constexpr int TAIL_ROWS = 8;
constexpr int MAXV = 4;

__global__ __launch_bounds__(256) void resblock_tail_kernel(
    const float* __restrict__ c2, const float* __restrict__ partials, const float* __restrict__ gamma,
    const float* __restrict__ beta, const int32_t* __restrict__ lens, const float* __restrict__ rres,
    const float* __restrict__ scale, const float* __restrict__ shift, int64_t ld_ss, float* __restrict__ y, int T,
    int C, int groups, float gn_eps, float ln_eps, int valid_stats, const float* __restrict__ ln2_gamma,
    const float* __restrict__ ln2_beta, float* __restrict__ y2, float ln2_eps, int rows_per_wg) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_wg;
  const int len = lens ? min(lens[b], T) : T;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4n = C / 4;
  const int cpg = C / groups;
  const float inv_c = 1.0f / (float)C;
  const int r_end = min(r0 + rows_per_wg, T);
  // a wave's row is on its way while the statistics are reduced (one memory round trip instead of two in a row)
  float4 xin[MAXV], rin[MAXV];
  auto fetch = [&](const int r) {
    const int64_t row = ((int64_t)b * T + r) * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      xin[i] = rin[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < c4n) {
        rin[i] = *reinterpret_cast<const float4*>(rres + row + c4 * 4);
        if (r < len) xin[i] = *reinterpret_cast<const float4*>(c2 + row + c4 * 4);
      }
    }
  };
  if (r0 + wave < r_end) fetch(r0 + wave);
  group_stats(partials, b, T, C, groups, gn_eps, s_mean, s_rstd, valid_stats ? len : T);
  for (int r = r0 + wave; r < r_end; r += 4) {
    if (r != r0 + wave) fetch(r);
    const int64_t row = ((int64_t)b * T + r) * C;
    const bool valid = r < len;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < c4n) {
        const int c = c4 * 4;
        const float4 rr = rin[i];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
          const float4 x = xin[i];
          const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
          const float4 be = *reinterpret_cast<const float4*>(beta + c);
          const int g = c / cpg;
          const float m = s_mean[g], rs = s_rstd[g];
          o.x = srn_mish((x.x - m) * rs * ga.x + be.x);
          o.y = srn_mish((x.y - m) * rs * ga.y + be.y);
          o.z = srn_mish((x.z - m) * rs * ga.z + be.z);
          o.w = srn_mish((x.w - m) * rs * ga.w + be.w);
        }
        v[i] = make_float4(o.x + rr.x, o.y + rr.y, o.z + rr.z, o.w + rr.w);
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
    const float stdv = sqrtf(wave_sum(sq) * inv_c + ln_eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < c4n) {
        const int c = c4 * 4;
        const float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)b * ld_ss + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + (int64_t)b * ld_ss + c);
        float4 o;
        o.x = (v[i].x - mean) / stdv * sc.x + sh.x;
        o.y = (v[i].y - mean) / stdv * sc.y + sh.y;
        o.z = (v[i].z - mean) / stdv * sc.z + sh.z;
        o.w = (v[i].w - mean) / stdv * sc.w + sh.w;
        *reinterpret_cast<float4*>(y + row + c) = o;
        v[i] = o;
      }
    }
    if (y2 != nullptr) {
      // the transformer block's first LayerNorm (transformer.py:286) of the row just produced, from registers:
      // layernorm_kernel's arithmetic, operation for operation
      float sum2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < c4n) sum2 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      const float mean2 = wave_sum(sum2) * inv_c;
      float sq2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        if (lane + 64 * i < c4n) {
          const float dx = v[i].x - mean2, dy = v[i].y - mean2, dz = v[i].z - mean2, dw = v[i].w - mean2;
          sq2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
      }
      const float rstd2 = 1.0f / sqrtf(wave_sum(sq2) * inv_c + ln2_eps);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < c4n) {
          const int c = c4 * 4;
          const float4 ga = *reinterpret_cast<const float4*>(ln2_gamma + c);
          const float4 be = *reinterpret_cast<const float4*>(ln2_beta + c);
          float4 o;
          o.x = (v[i].x - mean2) * rstd2 * ga.x + be.x;
          o.y = (v[i].y - mean2) * rstd2 * ga.y + be.y;
          o.z = (v[i].z - mean2) * rstd2 * ga.z + be.z;
          o.w = (v[i].w - mean2) * rstd2 * ga.w + be.w;
          *reinterpret_cast<float4*>(y2 + row + c) = o;
        }
      }
    }
  }
}

// nn.LayerNorm: one wavefront per row.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int64_t rows, int C, float eps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4n = C / 4;
  const float inv_c = 1.0f / (float)C;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t row = r * C;
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
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * ga.x + be.x;
        o.y = (v[i].y - mean) * rstd * ga.y + be.y;
        o.z = (v[i].z - mean) * rstd * ga.z + be.z;
        o.w = (v[i].w - mean) * rstd * ga.w + be.w;
        *reinterpret_cast<float4*>(y + row + c) = o;
      }
    }
  }
}

// Masked row softmax, in place, one wavefront per row, row held in registers (NV float4 per lane).
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, const int32_t* __restrict__ lens,
                                                           int64_t n_rows, int n_head, int L, int ld) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= n_rows) return;
  const int z = (int)(r / L);
  int len = L;
  if (lens) len = min(lens[z / n_head], L);
  float* row = s + r * ld;
  const int ld4 = ld / 4;
  float4 v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (c4 < ld4) {
      const float4 t = *reinterpret_cast<const float4*>(row + c4 * 4);
      const int c = c4 * 4;
      if (c + 0 < len) v[i].x = t.x;
      if (c + 1 < len) v[i].y = t.y;
      if (c + 2 < len) v[i].z = t.z;
      if (c + 3 < len) v[i].w = t.w;
      mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (lane + 64 * i < ld4) {
      v[i].x = expf(v[i].x - mx);
      v[i].y = expf(v[i].y - mx);
      v[i].z = expf(v[i].z - mx);
      v[i].w = expf(v[i].w - mx);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < ld4) {
      float4 o;
      o.x = v[i].x / sum;
      o.y = v[i].y / sum;
      o.z = v[i].z / sum;
      o.w = v[i].w / sum;
      *reinterpret_cast<float4*>(row + c4 * 4) = o;
    }
  }
}

__global__ void sinusoidal_emb_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int dim,
                                      int ld, float scale) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  const int i = idx / half, k = idx - i * half;
  // decoder.py:58-62: emb = log(10000) / (half - 1); f_k = exp(k * -emb); arg = scale * t * f_k
  const float emb = logf(10000.0f) / (float)(half - 1);
  const float f = expf((float)k * -emb);
  const float arg = scale * t[i] * f;
  out[(int64_t)i * ld + k] = sinf(arg);
  out[(int64_t)i * ld + half + k] = cosf(arg);
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int64_t src_bs, int ld_src, int sc0,
                                     float* __restrict__ dst, int64_t dst_bs, int ld_dst, int dc0, int T, int C) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)T * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx / C), c = (int)(idx - (int64_t)t * C);
    dst[(int64_t)b * dst_bs + (int64_t)t * ld_dst + dc0 + c] = src[(int64_t)b * src_bs + (int64_t)t * ld_src + sc0 + c];
  }
}

// ragged time-concat: rows [0, n_rows[b]) of item b go to dst rows row_off[b] + r (channels dc0 .. dc0 + C)
__global__ void scatter_rows_kernel(const float* __restrict__ src, int64_t src_bs, int ld_src,
                                    float* __restrict__ dst, int64_t dst_bs, int ld_dst, int dc0,
                                    const int32_t* __restrict__ row_off, const int32_t* __restrict__ n_rows, int T,
                                    int C) {
  const int b = blockIdx.y;
  const int off = row_off ? row_off[b] : 0;
  const int64_t total = (int64_t)min(n_rows ? n_rows[b] : T, T) * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx / C), c = (int)(idx - (int64_t)t * C);
    dst[(int64_t)b * dst_bs + (int64_t)(off + t) * ld_dst + dc0 + c] = src[(int64_t)b * src_bs + (int64_t)t * ld_src + c];
  }
}

// dst[b][c][r] = src[b][r][c]  (32x32 LDS tile transpose)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                        int Cc, int64_t src_bs, int ld_src, int64_t dst_bs,
                                                        int ld_dst) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? src[(int64_t)b * src_bs + (int64_t)r * ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) dst[(int64_t)b * dst_bs + (int64_t)c * ld_dst + r] = tile[tx][i];
  }
}

// the same for a table of problems: a block finds its entry by the prefix of block counts, then its (b, row tile,
// column tile).  Entries with a narrow side (a conv's k = 3 taps) would waste 29 of 32 tile columns: they go element
// by element instead (1024 dst elements per block, reads strided through L2).
__global__ __launch_bounds__(256) void transpose_multi_kernel(const SrnTransposeList L) {
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < L.n && (int)blockIdx.x >= L.first_block[e + 1]) ++e;
  const int blk = blockIdx.x - L.first_block[e];
  const float* __restrict__ src = L.src[e];
  float* __restrict__ dst = L.dst[e];
  const int R = L.R[e], Cc = L.Cc[e], ld_src = L.ld_src[e], ld_dst = L.ld_dst[e];
  const int64_t src_bs = L.src_bs[e], dst_bs = L.dst_bs[e];
  if (R < 16 || Cc < 16) {
    const int64_t per_b = (int64_t)R * Cc, total = per_b * L.B[e];
    for (int i = 0; i < 4; ++i) {
      const int64_t idx = (int64_t)blk * 1024 + i * 256 + threadIdx.x;  // dst order: b, c, r
      if (idx >= total) return;
      const int b = (int)(idx / per_b);
      const int rem = (int)(idx - (int64_t)b * per_b);
      const int c = rem / R, r = rem - c * R;
      dst[b * dst_bs + (int64_t)c * ld_dst + r] = src[b * src_bs + (int64_t)r * ld_src + c];
    }
    return;
  }
  const int tc = (Cc + 31) / 32, tr = (R + 31) / 32;
  const int b = blk / (tc * tr), t = blk - b * tc * tr;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? src[b * src_bs + (int64_t)r * ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) dst[b * dst_bs + (int64_t)c * ld_dst + r] = tile[tx][i];
  }
}

__global__ void renorm_kernel(const float* __restrict__ x, const float* __restrict__ ts, const float* __restrict__ tm,
                              const float* __restrict__ vm, const float* __restrict__ vs, float* __restrict__ y,
                              int64_t total, int C) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    float v = x[idx];
    if (ts) v = v * ts[c] + tm[c];  // vocoder.py:54
    y[idx] = (v - vm[c]) / vs[c];   // vocoder.py:56
  }
}

// HiFi-GAN output stage: LeakyReLU -> Conv1d(C -> 1, k) -> tanh.  x (B, T, C) channels-last, w (k, C), y (B, T).
// HBM-bound (one read of x): a workgroup stages the OC_BT + K - 1 rows its 256 outputs need in LDS with coalesced
// 16-B loads (LeakyReLU applied on the way in; 144-B row pitch, so the per-lane ds_read_b128 of 16 consecutive rows
// is conflict-free) and every lane then forms one output from LDS.  The round-1 kernel read its 7 x 128 B per lane
// straight from global memory, 64 different lines per load instruction (1.2 TB/s).
constexpr int OC_BT = 256;
template <int K, int CC>
__global__ __launch_bounds__(256) void out_conv_tanh_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int T, float slope) {
  constexpr int PITCH = CC + 4;  // floats
  constexpr int ROWS = OC_BT + K - 1;
  __shared__ __attribute__((aligned(16))) float sx[ROWS * PITCH];
  __shared__ __attribute__((aligned(16))) float sw[K * CC];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * OC_BT;
  const float* xb = x + (int64_t)b * T * CC;
  for (int i = threadIdx.x; i < K * CC; i += 256) sw[i] = w[i];
  constexpr int C4 = CC / 4;
  for (int i = threadIdx.x; i < ROWS * C4; i += 256) {
    const int r = i / C4, c4 = i - r * C4;
    const int ti = t0 + r - (K - 1) / 2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ti >= 0 && ti < T) v = leaky4(*reinterpret_cast<const float4*>(xb + (int64_t)ti * CC + c4 * 4), slope);
    *reinterpret_cast<float4*>(sx + r * PITCH + c4 * 4) = v;
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= T) return;
  float acc = bias[0];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const float* row = sx + (threadIdx.x + j) * PITCH;
#pragma unroll
    for (int c4 = 0; c4 < C4; ++c4) {
      const float4 v = *reinterpret_cast<const float4*>(row + c4 * 4);
      const float4 ww = *reinterpret_cast<const float4*>(sw + j * CC + c4 * 4);
      acc = fmaf(v.x, ww.x, acc);
      acc = fmaf(v.y, ww.y, acc);
      acc = fmaf(v.z, ww.z, acc);
      acc = fmaf(v.w, ww.w, acc);
    }
  }
  y[(int64_t)b * T + t] = tanhf(acc);
}

// generic (any C % 4 == 0, k <= 16) fallback of the above
__global__ __launch_bounds__(256) void out_conv_tanh_generic(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int T, int C, int K, float slope) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* xb = x + (int64_t)b * T * C;
  float acc = bias[0];
  for (int j = 0; j < K; ++j) {
    const int ti = t + j - (K - 1) / 2;
    if (ti < 0 || ti >= T) continue;
    for (int c = 0; c < C; ++c) {
      float v = xb[(int64_t)ti * C + c];
      v = v > 0.f ? v : v * slope;
      acc = fmaf(v, w[j * C + c], acc);
    }
  }
  y[(int64_t)b * T + t] = tanhf(acc);
}

}  // namespace

// =================================================================================================
extern "C" int srn_gn_mish_apply(const float* x, const float* gn_partials, const float* gamma, const float* beta,
                                 const float* time_bias, int64_t time_bias_bs, const int32_t* lens, float* y, int B,
                                 int T, int C, int groups, float eps, int valid_stats, void* stream) {
  SRN_CHECK_ARG(x && gn_partials && gamma && beta && y, "gn_mish_apply: null pointer");
  SRN_CHECK_ARG(!valid_stats || lens, "gn_mish_apply: valid_stats needs lens");
  SRN_CHECK_ARG(B > 0 && T > 0 && C > 0 && groups > 0 && groups <= 64 && C % groups == 0 && (C / groups) % 32 == 0,
                "gn_mish_apply: need (C / groups) %% 32 == 0 (C=%d groups=%d)", C, groups);
  const int rows_per_wg = (int64_t)B * ((T + GN_ROWS - 1) / GN_ROWS) < SMALL_GRID ? GN_ROWS / 2 : GN_ROWS;
  dim3 grid((T + rows_per_wg - 1) / rows_per_wg, B);
  hipLaunchKernelGGL(gn_mish_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, gn_partials, gamma, beta,
                     time_bias, lens, y, T, C, groups, eps, time_bias_bs, valid_stats, rows_per_wg);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_resblock_tail(const float* c2, const float* gn_partials, const float* gamma, const float* beta,
                                 const int32_t* lens, const float* r, const float* scale, const float* shift,
                                 int64_t ld_ss, float* y, int B, int T, int C, int groups, float gn_eps,
                                 float ln_eps, int valid_stats, void* stream) {
  SRN_CHECK_ARG(c2 && gn_partials && gamma && beta && r && scale && shift && y, "resblock_tail: null pointer");
  SRN_CHECK_ARG(!valid_stats || lens, "resblock_tail: valid_stats needs lens");
  SRN_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV, "resblock_tail: C=%d unsupported", C);
  SRN_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0 && (C / groups) % 32 == 0,
                "resblock_tail: need (C / groups) %% 32 == 0");
  const int rows_per_wg = (int64_t)B * ((T + TAIL_ROWS - 1) / TAIL_ROWS) < SMALL_GRID ? TAIL_ROWS / 2 : TAIL_ROWS;
  dim3 grid((T + rows_per_wg - 1) / rows_per_wg, B);
  hipLaunchKernelGGL(resblock_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, c2, gn_partials, gamma, beta,
                     lens, r, scale, shift, ld_ss, y, T, C, groups, gn_eps, ln_eps, valid_stats, nullptr, nullptr,
                     nullptr, 0.f, rows_per_wg);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_resblock_tail_ln(const float* c2, const float* gn_partials, const float* gamma, const float* beta,
                                    const int32_t* lens, const float* r, const float* scale, const float* shift,
                                    int64_t ld_ss, float* y, int B, int T, int C, int groups, float gn_eps,
                                    float ln_eps, int valid_stats, const float* ln2_gamma, const float* ln2_beta,
                                    float* y2, float ln2_eps, void* stream) {
  SRN_CHECK_ARG(c2 && gn_partials && gamma && beta && r && scale && shift && y && ln2_gamma && ln2_beta && y2,
                "resblock_tail_ln: null pointer");
  SRN_CHECK_ARG(!valid_stats || lens, "resblock_tail_ln: valid_stats needs lens");
  SRN_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV, "resblock_tail_ln: C=%d unsupported", C);
  SRN_CHECK_ARG(groups > 0 && groups <= 64 && C % groups == 0 && (C / groups) % 32 == 0,
                "resblock_tail_ln: need (C / groups) %% 32 == 0");
  const int rows_per_wg = (int64_t)B * ((T + TAIL_ROWS - 1) / TAIL_ROWS) < SMALL_GRID ? TAIL_ROWS / 2 : TAIL_ROWS;
  dim3 grid((T + rows_per_wg - 1) / rows_per_wg, B);
  hipLaunchKernelGGL(resblock_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, c2, gn_partials, gamma, beta,
                     lens, r, scale, shift, ld_ss, y, T, C, groups, gn_eps, ln_eps, valid_stats, ln2_gamma, ln2_beta,
                     y2, ln2_eps, rows_per_wg);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C,
                             float eps, void* stream) {
  SRN_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
  SRN_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 256 * MAXV, "layernorm: C=%d unsupported", C);
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     rows, C, eps);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_softmax_rows(float* s, const int32_t* lens, int Z, int n_head, int L, int ld, void* stream) {
  SRN_CHECK_ARG(s != nullptr && Z > 0 && n_head > 0 && L > 0 && ld >= L && ld % 4 == 0, "softmax_rows: bad args");
  const int64_t rows = (int64_t)Z * L;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  const int nv = (ld / 4 + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  if (nv <= 2) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else if (nv <= 4) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else if (nv <= 8) hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else if (nv <= 12) hipLaunchKernelGGL(softmax_rows_kernel<12>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else if (nv <= 20) hipLaunchKernelGGL(softmax_rows_kernel<20>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else if (nv <= 36) hipLaunchKernelGGL(softmax_rows_kernel<36>, dim3(blocks), dim3(256), 0, st, s, lens, rows, n_head, L, ld);
  else {
    srn_set_error("softmax_rows: L=%d too long (max 9216)", L);
    return -1;
  }
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_sinusoidal_emb(const float* t, float* out, int n, int dim, int ld, float scale, void* stream) {
  SRN_CHECK_ARG(t && out && n > 0 && dim > 2 && dim % 2 == 0 && ld >= dim, "sinusoidal_emb: bad args");
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(sinusoidal_emb_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, n,
                     dim, ld, scale);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_copy_channels(const float* src, int64_t src_bs, int ld_src, int sc0, float* dst, int64_t dst_bs,
                                 int ld_dst, int dc0, int B, int T, int C, void* stream) {
  SRN_CHECK_ARG(src && dst && B > 0 && T > 0 && C > 0, "copy_channels: bad args");
  int64_t blocks = ((int64_t)T * C + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy_channels_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, src, src_bs,
                     ld_src, sc0, dst, dst_bs, ld_dst, dc0, T, C);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_scatter_rows(const float* src, int64_t src_bs, int ld_src, float* dst, int64_t dst_bs, int ld_dst,
                                int dc0, const int32_t* row_off, const int32_t* n_rows, int B, int T, int C,
                                void* stream) {
  SRN_CHECK_ARG(src && dst && B > 0 && T > 0 && C > 0, "scatter_rows: bad args");
  int64_t blocks = ((int64_t)T * C + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, src, src_bs,
                     ld_src, dst, dst_bs, ld_dst, dc0, row_off, n_rows, T, C);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_transpose_ct(const float* src, float* dst, int B, int R, int Cc, int64_t src_bs, int ld_src,
                                int64_t dst_bs, int ld_dst, void* stream) {
  SRN_CHECK_ARG(src && dst && B > 0 && R > 0 && Cc > 0, "transpose_ct: bad args");
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, B);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, R, Cc, src_bs, ld_src,
                     dst_bs, ld_dst);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_transpose_multi(const SrnTransposeList* list, void* stream) {
  SRN_CHECK_ARG(list != nullptr && list->n > 0 && list->n <= SRN_TR_LIST_MAX, "transpose_multi: bad list");
  SrnTransposeList L = *list;
  int64_t blocks = 0;
  for (int e = 0; e < L.n; ++e) {
    SRN_CHECK_ARG(L.src[e] && L.dst[e] && L.B[e] > 0 && L.R[e] > 0 && L.Cc[e] > 0 && L.ld_src[e] >= L.Cc[e] &&
                      L.ld_dst[e] >= L.R[e],
                  "transpose_multi: entry %d: bad pointers / sizes / leading dimensions", e);
    L.first_block[e] = (int32_t)blocks;
    if (L.R[e] < 16 || L.Cc[e] < 16) blocks += ((int64_t)L.B[e] * L.R[e] * L.Cc[e] + 1023) / 1024;
    else blocks += (int64_t)L.B[e] * ((L.R[e] + 31) / 32) * ((L.Cc[e] + 31) / 32);
    SRN_CHECK_ARG(blocks < (1ll << 31), "transpose_multi: grid too large");
  }
  L.first_block[L.n] = (int32_t)blocks;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_renorm(const float* x, const float* trg_scale, const float* trg_mean, const float* voc_mean,
                          const float* voc_scale, float* y, int64_t rows, int C, void* stream) {
  SRN_CHECK_ARG(x && voc_mean && voc_scale && y && rows > 0 && C > 0, "renorm: bad args");
  SRN_CHECK_ARG((trg_scale == nullptr) == (trg_mean == nullptr), "renorm: trg_scale/trg_mean must come together");
  int64_t blocks = (rows * C + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(renorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, trg_scale, trg_mean,
                     voc_mean, voc_scale, y, rows * C, C);
  SRN_CHECK_LAUNCH();
  return 0;
}

extern "C" int srn_out_conv_tanh(const float* x, const float* w, const float* bias, float* y, int B, int T, int C,
                                 int k, float slope, void* stream) {
  SRN_CHECK_ARG(x && w && bias && y && B > 0 && T > 0 && C > 0 && k > 0 && k % 2 == 1 && k <= 16,
                "out_conv_tanh: bad args");
  dim3 grid((T + 255) / 256, B);
  if (C == 32 && k == 7 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    hipLaunchKernelGGL((out_conv_tanh_kernel<7, 32>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, T, slope);
  } else {
    hipLaunchKernelGGL(out_conv_tanh_generic, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, T, C, k, slope);
  }
  SRN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SiFiGAN "pitch-dependent dilated conv" operand gather (row a9, parity unpinned: see oracle/sifigan_oracle.py):
//   out[b, t, 0:C]   = lrelu(x[b, t])
//   out[b, t, C:2C]  = lrelu(x[b, t - r]),  out[b, t, 2C:3C] = lrelu(x[b, t + r]),   r = rint(d[b, t] * dilation)
// (zero outside the signal).  Whole 128-B+ channel rows move as float4; one wave handles several rows.
namespace {
__global__ __launch_bounds__(256) void pd_gather_kernel(const float* __restrict__ x, const float* __restrict__ d,
                                                        float* __restrict__ out, int T, int C, float dilation,
                                                        float slope) {
  const int b = blockIdx.y;
  const int c4n = C / 4;
  const int64_t total = (int64_t)T * c4n;
  const float* xb = x + (int64_t)b * T * C;
  float* ob = out + (int64_t)b * T * 3 * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int t = (int)(idx / c4n);
    const int c = (int)(idx - (int64_t)t * c4n) * 4;
    const int r = (int)rintf(d[(int64_t)b * T + t] * dilation);
    const int tp = t - r, tf = t + r;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v0 = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + c);
    float4 vp = (tp >= 0 && tp < T) ? *reinterpret_cast<const float4*>(xb + (int64_t)tp * C + c) : z;
    float4 vf = (tf >= 0 && tf < T) ? *reinterpret_cast<const float4*>(xb + (int64_t)tf * C + c) : z;
    auto lr = [slope](float4 v) {
      v.x = v.x > 0.f ? v.x : v.x * slope;
      v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope;
      v.w = v.w > 0.f ? v.w : v.w * slope;
      return v;
    };
    float* o = ob + (int64_t)t * 3 * C + c;
    *reinterpret_cast<float4*>(o) = lr(v0);
    *reinterpret_cast<float4*>(o + C) = lr(vp);
    *reinterpret_cast<float4*>(o + 2 * C) = lr(vf);
  }
}
}  // namespace

extern "C" int srn_pd_gather(const float* x, const float* d, float* out, int B, int T, int C, float dilation,
                             float slope, void* stream) {
  SRN_CHECK_ARG(x && d && out && B > 0 && T > 0 && C > 0 && C % 4 == 0, "pd_gather: bad args");
  int64_t blocks = ((int64_t)T * (C / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pd_gather_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, x, d, out, T, C,
                     dilation, slope);
  SRN_CHECK_LAUNCH();
  return 0;
}
