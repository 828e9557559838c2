// conv_common.h — pieces shared by the implicit-GEMM kernels (conv_gemm.hip, conv_halo.hip): fragment types,
// the fp32 -> (hi, lo) bf16 split, the swizzled bf16 LDS image, the XCD-aware block map and the fused epilogue.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// component-wise select (a float4 ?: is lowered through scratch memory by hipcc)
__device__ __forceinline__ float4 sel4(unsigned ok, const float4& v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

__device__ __forceinline__ float4 leaky4(float4 v, float slope) {
  v.x = v.x > 0.f ? v.x : v.x * slope;
  v.y = v.y > 0.f ? v.y : v.y * slope;
  v.z = v.z > 0.f ? v.z : v.z * slope;
  v.w = v.w > 0.f ? v.w : v.w * slope;
  return v;
}

// fp32 -> (hi, lo) bf16 pair with x ~= hi + lo to 2^-17 relative (round-to-nearest twice; hipcc emits
// v_cvt_pk_bf16_f32 + shift/and + v_sub: 3 VALU ops per element)
__device__ __forceinline__ void split4(const float4& v, bf16x4& hi, bf16x4& lo) {
  hi[0] = (__bf16)v.x;
  hi[1] = (__bf16)v.y;
  hi[2] = (__bf16)v.z;
  hi[3] = (__bf16)v.w;
  lo[0] = (__bf16)(v.x - (float)hi[0]);
  lo[1] = (__bf16)(v.y - (float)hi[1]);
  lo[2] = (__bf16)(v.z - (float)hi[2]);
  lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// byte offset of bf16 element (row, k) in a [rows][32] bf16 tile with 64-B rows whose four 16-B chunks are
// XOR-swizzled by (row >> 2) & 3: ds_read_b128 of 16 different rows at one logical chunk is conflict-free
__device__ __forceinline__ int bf_off(int row, int k) {
  return row * 64 + ((((k >> 3) ^ (row >> 2)) & 3) << 4) + ((k & 7) << 1);
}

// XCD-aware, bijective blockIdx -> logical tile id (blocks b, b+8, ... share an XCD's L2): consecutive logical
// ids (which share the A tile / neighbouring weight tiles) are dealt to one XCD.
__device__ __forceinline__ int xcd_logical_block() {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// logical tile id -> (z, m tile, n tile). Inside one z the tiles are walked in bands of TILE_BAND m-tiles, m
// fastest: the ~64 blocks resident on one XCD then cover an 8 x 8 patch (16 operand panels for 64 tiles) instead
// of one m-tile row x all n-tiles (n_tiles + 1 panels), which is what keeps wide-N weights inside the 4 MB L2.
constexpr int TILE_BAND = 8;
__device__ __forceinline__ void tile_coords(const int logical, const int m_tiles, const int n_tiles, int& z,
                                            int& mt_i, int& nt_i) {
  const int per_z = m_tiles * n_tiles;
  z = logical / per_z;
  const int l = logical - z * per_z;
  const int band = l / (TILE_BAND * n_tiles);
  const int m0 = band * TILE_BAND;
  const int gm = min(TILE_BAND, m_tiles - m0);
  const int r = l - band * TILE_BAND * n_tiles;
  nt_i = r / gm;
  mt_i = m0 + (r - nt_i * gm);
}

// Fused epilogue of one wave's MT x NT accumulator tiles (C/D map of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): alpha, bias, GEGLU gate, output mask, residual add / axpy,
// second residual, post op, strided store, GroupNorm partial sums.
// No __restrict__: res / res2 may alias out (in-place Euler update, HiFi-GAN stage sum).
//
// Addressing: every tensor is walked as  wave-uniform 64-bit base (SGPRs, advanced per accumulator row)  +  one
// per-lane 32-bit element offset that does not depend on the row, so an element costs one store (and one load per
// residual) plus scalar adds.  GEGLU / RES are compile-time: the common plain epilogue carries no residual or
// post-op code, and interior 32x32 sub-tiles skip the bounds predicates.
template <int MT, int NT, bool GEGLU, bool RES>
__device__ __forceinline__ void conv_epilogue_impl(const SrnConvParams& p, f32x16 (&acc)[MT][NT], const int zb,
                                                   const int zh, const int t0, const int n0, const int wm0,
                                                   const int wn0, const int lane) {
  const int li = lane & 31;
  const int lh = lane >> 5;
  float* out = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs;
  const float* res = RES && p.res ? p.res + (int64_t)zb * p.res_bs + (int64_t)zh * p.res_hs : nullptr;
  const float* res2 = RES && p.res2 ? p.res2 + (int64_t)zb * p.res2_bs : nullptr;
  int len_out = p.T_out;
  if (p.len_out) len_out = min(p.len_out[zb], p.T_out);
  const int gn_mt = (p.T_out + 31) / 32;
  const int gn_nt = p.N / 32;
  const int ts = p.out_t_stride;
  const float alpha = p.alpha;
  const int res_mode = RES ? p.res_mode : SRN_RES_NONE;
  const int post = RES ? p.post : SRN_POST_NONE;
  // bias of every column block up front: `bias` may alias `out` as far as the compiler knows, so a load inside the
  // sub-tile loop cannot move above the previous sub-tile's stores and each sub-tile would start with a round trip
  float bias_all[NT], bias_gate[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ncol = n0 + wn0 + n * 32 + li;
    const bool okc = p.bias != nullptr && ncol < p.N;
    bias_all[n] = okc ? p.bias[ncol] : 0.f;
    bias_gate[n] = (GEGLU && okc && ncol + 32 < p.N) ? p.bias[ncol + 32] : 0.f;
  }

#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tm = t0 + wm0 + m * 32;  // wave-uniform first row of the sub-tile
    if (tm >= p.T_out) continue;
    const int64_t orow0 = (int64_t)tm * ts + p.out_t_off;
    const bool rows_in = tm + 32 <= p.T_out;
    const bool rows_live = tm + 32 <= len_out;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (GEGLU && (n & 1)) continue;  // gate tiles are consumed with their value tile
      const int nc0 = n0 + wn0 + n * 32;  // wave-uniform first GEMM column
      const int oc0 = GEGLU ? (nc0 >> 6) * 32 : nc0;
      const int ncol = nc0 + li;
      const int ocol = oc0 + li;
      const float bias_v = bias_all[n], bias_g = bias_gate[n];
      bool col_ok = ncol < p.N;
      col_ok = col_ok && ocol < p.N_out;
      if constexpr (!GEGLU && !RES) {
        if (p.out_tr != nullptr && nc0 >= p.out_tr_col0) {
          // transposed tail (V^T of the QKV projection): the lane's 4 consecutive rows of one column are 16
          // contiguous bytes of out_tr[zb][col][t]
          float* t_u = p.out_tr + (int64_t)zb * p.out_tr_bs + (int64_t)(nc0 - p.out_tr_col0) * p.ld_out_tr + tm;
          const unsigned t_v = (unsigned)li * (unsigned)p.ld_out_tr + 4 * lh;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int trow = tm + 8 * gq + 4 * lh;
            float4 v4;
            v4.x = (rows_live || trow + 0 < len_out) ? acc[m][n][4 * gq + 0] * alpha + bias_v : 0.f;
            v4.y = (rows_live || trow + 1 < len_out) ? acc[m][n][4 * gq + 1] * alpha + bias_v : 0.f;
            v4.z = (rows_live || trow + 2 < len_out) ? acc[m][n][4 * gq + 2] * alpha + bias_v : 0.f;
            v4.w = (rows_live || trow + 3 < len_out) ? acc[m][n][4 * gq + 3] * alpha + bias_v : 0.f;
            float* dst = t_u + 8 * gq + t_v;
            if (col_ok && trow + 3 < p.T_out) {
              *reinterpret_cast<float4*>(dst) = v4;
            } else if (col_ok) {
              if (trow + 0 < p.T_out) dst[0] = v4.x;
              if (trow + 1 < p.T_out) dst[1] = v4.y;
              if (trow + 2 < p.T_out) dst[2] = v4.z;
            }
          }
          continue;
        }
      }
      const bool interior = rows_in && nc0 + 32 <= p.N && oc0 + 32 <= p.N_out;
      // wave-uniform bases at sub-tile row 0 / column 0 and the per-lane, row-independent offsets
      float* o_u = out + orow0 * p.ld_out + oc0;
      const unsigned o_v = (unsigned)(4 * lh * ts) * (unsigned)p.ld_out + li;
      const int64_t o_rs = (int64_t)ts * p.ld_out;
      const float* r_u = res ? res + orow0 * p.ld_res + oc0 : nullptr;
      const unsigned r_v = (unsigned)(4 * lh * ts) * (unsigned)p.ld_res + li;
      const int64_t r_rs = (int64_t)ts * p.ld_res;
      const float* q_u = res2 ? res2 + orow0 * p.ld_res2 + oc0 : nullptr;
      const unsigned q_v = (unsigned)(4 * lh * ts) * (unsigned)p.ld_res2 + li;
      const int64_t q_rs = (int64_t)ts * p.ld_res2;
      float s1 = 0.f, s2 = 0.f;
      // Residuals of the whole sub-tile first, back to back: `res` / `res2` may alias `out`, so the compiler cannot
      // move a residual load above an earlier store and a load-per-row loop pays one memory round trip per row
      // (16 per sub-tile).  Reading before writing is safe for the supported aliasing: a lane reads exactly the
      // elements it later writes.
      float rv[16], qv[16];
      if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const bool ok = interior || (col_ok && tm + dr + 4 * lh < p.T_out);
          rv[r] = (r_u && ok) ? (r_u + dr * r_rs)[r_v] : 0.f;
          qv[r] = (q_u && ok) ? (q_u + dr * q_rs)[q_v] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);  // compile-time row of this register (+ 4 lh per lane)
        float v = acc[m][n][r] * alpha + bias_v;
        if constexpr (GEGLU) {
          const float g = acc[m][(n + 1) % NT][r] * alpha + bias_g;
          v = v * srn_gelu_erf(g);
        }
        const int trow = tm + dr + 4 * lh;
        if (!rows_live && trow >= len_out) v = 0.f;
        const bool ok = interior || (col_ok && trow < p.T_out);
        if (ok) {
          if constexpr (RES) {
            if (res_mode == SRN_RES_ADD) v += rv[r];
            else if (res_mode == SRN_RES_AXPY) v = rv[r] + p.beta * v;
            if (q_u) v += qv[r];
            if (post == SRN_POST_DIV) v = v / p.post_div;
            else if (post == SRN_POST_TANH) v = tanhf(v);
            else if (post == SRN_POST_RELU) v = fmaxf(v, 0.f);
            else if (post == SRN_POST_LEAKY) v = v > 0.f ? v : v * p.post_div;
          }
          (o_u + dr * o_rs)[o_v] = v;
          s1 += v;
          s2 += v * v;
        }
      }
      if (p.gn_partials) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const int gmt = tm >> 5;
        const int gnt = nc0 >> 5;
        if (lane == 0 && gmt < gn_mt && gnt < gn_nt) {
          float* gp = p.gn_partials + (((int64_t)zb * gn_mt + gmt) * gn_nt + gnt) * 2;
          gp[0] = s1;
          gp[1] = s2;
        }
      }
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const SrnConvParams& p, f32x16 (&acc)[MT][NT], const int zb,
                                              const int zh, const int t0, const int n0, const int wm0_,
                                              const int wn0_, const int lane) {
  const int wm0 = __builtin_amdgcn_readfirstlane(wm0_);
  const int wn0 = __builtin_amdgcn_readfirstlane(wn0_);
  if constexpr (NT % 2 == 0) {
    if (p.geglu) {
      conv_epilogue_impl<MT, NT, true, false>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
      return;
    }
  }
  if (p.res_mode == SRN_RES_NONE && p.res2 == nullptr && p.post == SRN_POST_NONE)
    conv_epilogue_impl<MT, NT, false, false>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
  else
    conv_epilogue_impl<MT, NT, false, true>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

// split-K: raw partial sums of one wave's accumulator tiles to the workspace slab [slice][z][T_out][N]
template <int MT, int NT>
__device__ __forceinline__ void splitk_store(const SrnConvParams& p, f32x16 (&acc)[MT][NT], const int slice,
                                             const int z, const int t0, const int n0, const int wm0, const int wn0,
                                             const int lane) {
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int64_t Z = (int64_t)p.n_batch * p.n_head;
  float* ws = reinterpret_cast<float*>(p.ws) + (((int64_t)slice * Z + z) * p.T_out) * p.N;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n0 + wn0 + n * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = t0 + wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.T_out && col < p.N) ws[(int64_t)row * p.N + col] = acc[m][n][r];
      }
    }
}

// implemented in conv_halo.hip: receptive-field ("halo") variant for stride-1 multi-tap convs in split-bf16.
// Returns 1 if it handled the launch, 0 if the shape is not eligible, < 0 on error.
int srn_conv_halo_try(const SrnConvParams& p, int tile, hipStream_t stream);
// implemented in conv_fast.hip: lean split-bf16 kernel for C_in % 32 == 0, k-major weights.  Same return codes.
int srn_conv_fast_try(const SrnConvParams& p, int tile, hipStream_t stream, int ksplit = 1);
// implemented in conv_f32.hip: the exact-fp32 contraction with a VALU-free main loop (tile ids 7, 9, 10).  Same return codes.
int srn_conv_f32_try(const SrnConvParams& p, int tile, hipStream_t stream, int ksplit);
// implemented in conv_splitk.hip: K slices for launches that cannot fill the chip (1 = do not split), the workspace
// they need, and the reduction + epilogue over the partial sums.
int srn_splitk_plan(const SrnConvParams& p);
int64_t srn_splitk_bytes(const SrnConvParams& p, int ksplit);
int srn_splitk_reduce(const SrnConvParams& p, int ksplit, hipStream_t stream);
// implemented in conv_strip.hip: thin convs (C_in, N in {32, 64}) with the whole weight tensor LDS-resident.
int srn_conv_strip_try(const SrnConvParams& p, hipStream_t stream);
