// conv_f32.hip -- the exact-fp32 implicit-GEMM contraction of the path with a vector-ALU-free main loop.
//
// Why a separate kernel (round 4, tools/experiments/mfma_valu_mix.hip / mfma_valu_kinds.hip, profiles/r4_*):
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate because it runs on the SIMD's fp32 lanes -- a wave's ordinary
// VALU instructions are not "free beside the matrix pipe" as they are beside the bf16 MFMAs, they come straight out of
// the MFMA stream: every v_add / v_mov / v_fma issued between two fp32 MFMAs costs the SIMD 3-7 cycles of matrix time
// (64-bit adds and selects 5-7, v_exp 9-12), with 1-6 waves per SIMD alike; register-only MFMA + v_fma mixes top out
// at 0.90-0.94 of the MFMA-only rate in total FLOPs.  conv_fast.hip's fp32 instantiation spends ~200 VALU
// instructions per tile in its prologue (seven run-time integer divisions on the float unit, 64-bit row pointers),
// 6 per 32-deep step on pointer bumps and 24 more per step where a LeakyReLU rides in the prologue: 8-14 % of the
// matrix time at K = 512, where half of an Euler step lives.  Here:
//   * tile coordinates come from host-made magic multipliers: scalar ALU only, no division;
//   * operands are fetched with buffer loads: a wave-uniform descriptor per tensor, ONE 32-bit per-lane byte offset
//     per tile row (recomputed only when the tap or the concat half changes) and the walk along K in the
//     instruction's SCALAR offset -- the steady-state loop issues no VALU instruction at all (LeakyReLU: 2 per element);
//   * padded / masked rows carry an out-of-range offset: the range check returns zeros, no zero page, no selects;
//   * the last step's loads are peeled instead of parked.
// LDS image, fragment order and MFMA sequence are conv_fast.hip's single-stage fp32 form, so the 64 x 64 and 64 x 128
// tiles give bit-identical results to it.  New: a 32 x 64 tile whose two wave pairs each take half of every 32-deep
// step (partial accumulators joined through LDS in a fixed order) -- 5120 rows x 512 columns (the half-resolution
// levels) are 1280 such tiles = 5 per CU exactly, where 640 tiles of 64 x 64 leave CUs with 3 and with 2.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int BK = 32;
constexpr int ROW = 36;  // floats per LDS row: 32 + 4 pad (144 B: conflict-free ds_read_b128 of 16 rows)
constexpr unsigned OOB = 0x80000000u;  // >= every descriptor's num_records: the load returns zeros

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// n / d == (n * mul) >> shift for 0 <= n < 2^26 (host: make_fdiv); scalar operands stay on the scalar ALU
struct FDiv {
  uint32_t mul, shift;
};
__device__ __forceinline__ int fdiv(const int n, const FDiv d) {
  return (int)(((uint64_t)(uint32_t)n * d.mul) >> d.shift);
}

struct F32Launch {
  int m_tiles, n_tiles, ksplit, per_slice, tiles_all;
  FDiv d_per_z, d_band, d_head, d_tiles_all, d_cpt;
};

// KW: groups of waves that split every 32-deep step between them (each group owns the whole BM x BN tile)
template <int BM_, int BN_, int WM_, int WN_, int KW_, int MINW_, int PF_ = 1>
struct TCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, KW = KW_, MINW = MINW_, PF = PF_;
  static constexpr int MT = WM / 32, NT = WN / 32;
  static constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
  static_assert(WAVES_M * WAVES_N * KW == 4, "4 waves per workgroup");
  static constexpr int A_LD = BM / 32, B_LD = BN / 32;  // 16-B loads per thread per step
  static constexpr int KK = (BK / 8) / KW;               // 8-deep fragment groups per wave per step
  static_assert(KK * KW * 8 == BK, "the step divides over the wave groups");
  static constexpr int SMEM_BYTES = (BM + BN) * ROW * 4;
  static_assert((KW - 1) * WAVES_M * WAVES_N * MT * NT * 16 * 64 * 4 <= SMEM_BYTES, "join scratch fits the stage");
};

// Epilogue with the same contract as conv_common.h's conv_epilogue, written for the fp32 matrix pipe's economics: an
// interior 32 x 32 sub-tile (all rows < T_out and < len_out, all columns < N) costs one fma per element -- plus two
// per element where GroupNorm partial sums ride along, one add per residual -- and NO address arithmetic: stores and
// residual loads are buffer instructions with one per-lane offset per tensor and the row in the scalar offset.
// Edge sub-tiles, GEGLU and the transposed tail go through conv_epilogue_impl unchanged (same values bit for bit).
template <int MT, int NT>
__device__ __forceinline__ void f32_epilogue(const SrnConvParams& p, f32x16 (&acc)[MT][NT], const int zb, const int zh,
                                             const int t0, const int n0, const int wm0_, const int wn0_, const int lane) {
  const int wm0 = __builtin_amdgcn_readfirstlane(wm0_);
  const int wn0 = __builtin_amdgcn_readfirstlane(wn0_);
  if constexpr (NT % 2 == 0) {
    if (p.geglu) {
      conv_epilogue_impl<MT, NT, true, false>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
      return;
    }
  }
  const int li = lane & 31;
  const int lh = lane >> 5;
  int len_out = p.T_out;
  if (p.len_out) len_out = min(p.len_out[zb], p.T_out);
  const int ts = p.out_t_stride;
  const int rows_all = (p.T_out - 1) * ts + p.out_t_off;  // last output row that exists
  float* const out = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs;
  const __amdgpu_buffer_rsrc_t rs_o =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, (rows_all * p.ld_out + p.N_out) * 4, 0x00020000);
  const bool has_res = p.res_mode != SRN_RES_NONE;
  const float* const res = has_res ? p.res + (int64_t)zb * p.res_bs + (int64_t)zh * p.res_hs : out;
  const float* const res2 = p.res2 ? p.res2 + (int64_t)zb * p.res2_bs : out;
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(res), 0, has_res ? (rows_all * p.ld_res + p.N_out) * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(res2), 0, p.res2 ? (rows_all * p.ld_res2 + p.N_out) * 4 : 0, 0x00020000);
  const bool simple = !has_res && p.res2 == nullptr && p.post == SRN_POST_NONE;
  const float alpha = p.alpha;
  // per-lane byte offsets: row 4 lh of the sub-tile, column li
  const int o_v = (4 * lh * ts * p.ld_out + li) * 4;
  const int r_v = (4 * lh * ts * p.ld_res + li) * 4;
  const int q_v = (4 * lh * ts * p.ld_res2 + li) * 4;
  const int o_rs = ts * p.ld_out * 4, r_rs = ts * p.ld_res * 4, q_rs = ts * p.ld_res2 * 4;
  const int gn_mt = (p.T_out + 31) / 32, gn_nt = p.N / 32;
  const bool gn = p.gn_partials != nullptr;
  // bias of every column block up front (a load inside the sub-tile loop cannot move above the previous sub-tile's stores)
  float bias_all[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ncol = n0 + wn0 + n * 32 + li;
    bias_all[n] = (p.bias != nullptr && ncol < p.N) ? p.bias[ncol] : 0.f;
  }

#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tm = t0 + wm0 + m * 32;
    if (tm >= p.T_out) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int nc0 = n0 + wn0 + n * 32;
      const bool fast = tm + 32 <= len_out && nc0 + 32 <= p.N && nc0 + 32 <= p.N_out &&
                        !(p.out_tr != nullptr && nc0 >= p.out_tr_col0);
      if (!fast) {
        f32x16(&one)[1][1] = reinterpret_cast<f32x16(&)[1][1]>(acc[m][n]);
        if (simple) conv_epilogue_impl<1, 1, false, false>(p, one, zb, zh, t0, n0, wm0 + m * 32, wn0 + n * 32, lane);
        else conv_epilogue_impl<1, 1, false, true>(p, one, zb, zh, t0, n0, wm0 + m * 32, wn0 + n * 32, lane);
        continue;
      }
      const float bias = bias_all[n];
      const int row0 = tm * ts + p.out_t_off;
      const int o_s = (row0 * p.ld_out + nc0) * 4;
      float s1 = 0.f, s2 = 0.f;
      if (simple && !gn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const float v = acc[m][n][r] * alpha + bias;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, o_v, o_s + dr * o_rs, 0);
        }
      } else if (simple) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          const float v = acc[m][n][r] * alpha + bias;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, o_v, o_s + dr * o_rs, 0);
          s1 += v;
          s2 += v * v;
        }
      } else {
        // residuals of the whole sub-tile first: they may alias out (a lane reads exactly the elements it writes)
        const int r_s = (row0 * p.ld_res + nc0) * 4, q_s = (row0 * p.ld_res2 + nc0) * 4;
        float rv[16], qv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, r_v, r_s + dr * r_rs, 0));
          qv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_q, q_v, q_s + dr * q_rs, 0));
        }
        const int res_mode = p.res_mode, post = p.post;
        const bool has_q = p.res2 != nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          float v = acc[m][n][r] * alpha + bias;
          if (res_mode == SRN_RES_ADD) v += rv[r];
          else if (res_mode == SRN_RES_AXPY) v = rv[r] + p.beta * v;
          if (has_q) v += qv[r];
          if (post == SRN_POST_DIV) v = v / p.post_div;
          else if (post == SRN_POST_TANH) v = tanhf(v);
          else if (post == SRN_POST_RELU) v = fmaxf(v, 0.f);
          else if (post == SRN_POST_LEAKY) v = v > 0.f ? v : v * p.post_div;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, o_v, o_s + dr * o_rs, 0);
          s1 += v;
          s2 += v * v;
        }
      }
      if (gn) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const int gmt = tm >> 5, gnt = nc0 >> 5;
        if (lane == 0 && gmt < gn_mt && gnt < gn_nt) {
          float* gp = p.gn_partials + (((int64_t)zb * gn_mt + gmt) * gn_nt + gnt) * 2;
          gp[0] = s1;
          gp[1] = s2;
        }
      }
    }
  }
}

template <class C, int ACT>
__global__ __launch_bounds__(256, C::MINW) void conv_f32_kernel(const SrnConvParams p, const F32Launch L) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* const smem = reinterpret_cast<float*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which tile (scalar ALU only)
  int logical = xcd_logical_block();
  int slice = 0;
  if (L.ksplit > 1) {
    slice = fdiv(logical, L.d_tiles_all);
    logical -= slice * L.tiles_all;
  }
  const int per_z = L.m_tiles * L.n_tiles;
  const int z = fdiv(logical, L.d_per_z);
  const int l = logical - z * per_z;
  const int band = fdiv(l, L.d_band);  // bands of TILE_BAND m-tiles, m fastest (conv_common.h: tile_coords)
  const int m0 = band * TILE_BAND;
  const int gm = min(TILE_BAND, L.m_tiles - m0);
  const int r = l - band * TILE_BAND * L.n_tiles;
  int nt_i;
  switch (gm) {  // divisions by constants: scalar multiply-high
    case 8: nt_i = r >> 3; break;
    case 7: nt_i = (int)((unsigned)r / 7u); break;
    case 6: nt_i = (int)((unsigned)r / 6u); break;
    case 5: nt_i = (int)((unsigned)r / 5u); break;
    case 4: nt_i = r >> 2; break;
    case 3: nt_i = (int)((unsigned)r / 3u); break;
    case 2: nt_i = r >> 1; break;
    default: nt_i = r; break;
  }
  const int mt_i = m0 + (r - nt_i * gm);
  const int zb = fdiv(z, L.d_head);
  const int zh = z - zb * p.n_head;
  const int t0 = mt_i * BM;
  const int n0 = nt_i * BN;

  const int T_in = p.T_in;
  int len_in = T_in;
  if (p.len_in) len_in = min(p.len_in[zb], T_in);

  const int cpt = p.C_in >> 5;   // 32-channel steps per tap
  const int cp0 = p.C_in0 >> 5;  // ... of which from in0
  const int steps_all = p.n_taps * cpt;
  int s_begin = 0, n_steps = steps_all;
  if (L.ksplit > 1) {
    s_begin = slice * L.per_slice;
    n_steps = max(0, min(steps_all, s_begin + L.per_slice) - s_begin);
  }

  // ---- buffer descriptors (wave-uniform: kernel arguments and block coordinates only)
  const float* a0 = p.in0 + (int64_t)zb * p.in0_bs + (int64_t)zh * p.in0_hs;
  const float* a1 = p.in1 ? p.in1 + (int64_t)zb * p.in1_bs : a0;
  const float* wb = p.w + (int64_t)zb * p.w_bs + (int64_t)zh * p.w_hs;
  const int bytes0 = ((T_in - 1) * p.ld_in0 + p.C_in0) * 4;
  const int bytes1 = p.in1 ? ((T_in - 1) * p.ld_in1 + (p.C_in - p.C_in0)) * 4 : bytes0;
  const int bytesw = ((p.N - 1) * p.ldw + steps_all * BK) * 4;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a0), 0, bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1), 0, bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb), 0, bytesw, 0x00020000);

  // ---- per-lane row offsets
  const int c16 = (tid & 7) * 16;  // byte of this thread's 16-B piece inside the 128-B line
  const int lrow = tid >> 3;       // tile rows lrow + 32 i
  int a_tb[C::A_LD];               // input row of the tile row before the tap offset
#pragma unroll
  for (int i = 0; i < C::A_LD; ++i) a_tb[i] = min(t0 + lrow + 32 * i, p.T_out - 1) * p.in_stride;
  unsigned voff_b[C::B_LD];
#pragma unroll
  for (int i = 0; i < C::B_LD; ++i) voff_b[i] = __umul24((unsigned)min(n0 + lrow + 32 * i, p.N - 1), (unsigned)(p.ldw * 4)) + (unsigned)c16;
  unsigned voff_a[C::A_LD];
  int cur_tap, cur_seg, left, soff_a;
  int soff_b = s_begin * (BK * 4);
  auto a_setup = [&](const int tap, const int seg) {
    const int toff = p.tap_off[tap];
    const int ld4 = (seg == 0 ? p.ld_in0 : p.ld_in1) * 4;
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      int ti = a_tb[i] + toff;
      if (p.pad_reflect) {  // 2: mirror at the item's own end (ragged batches), else at the tensor's end
        const int T_ref = p.pad_reflect == 2 ? len_in : T_in;
        if (ti < 0) ti = -ti;
        if (ti >= T_ref) ti = 2 * (T_ref - 1) - ti;
      }
      voff_a[i] = (unsigned)ti < (unsigned)len_in ? __umul24((unsigned)ti, (unsigned)ld4) + (unsigned)c16 : OOB;  // T_in, 4 ld < 2^24 (host-checked)
    }
  };
  {
    cur_tap = L.ksplit > 1 ? fdiv(s_begin, L.d_cpt) : 0;
    const int within = s_begin - cur_tap * cpt;
    cur_seg = within >= cp0 ? 1 : 0;
    const int chunk = cur_seg ? within - cp0 : within;
    a_setup(min(cur_tap, p.n_taps - 1), cur_seg);
    left = (cur_seg ? cpt - cp0 : cp0) - chunk;
    soff_a = chunk * (BK * 4);
  }

  struct Regs {
    u32x4 pa[C::A_LD];
    u32x4 pb[C::B_LD];
  };
  auto load_issue = [&](Regs& R) {
    if (cur_seg == 0) {
#pragma unroll
      for (int i = 0; i < C::A_LD; ++i) R.pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs0, voff_a[i], soff_a, 0);
    } else {
#pragma unroll
      for (int i = 0; i < C::A_LD; ++i) R.pa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs1, voff_a[i], soff_a, 0);
    }
#pragma unroll
    for (int i = 0; i < C::B_LD; ++i) R.pb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voff_b[i], soff_b, 0);
    soff_a += BK * 4;
    soff_b += BK * 4;
  };
  auto advance = [&]() {  // wave-uniform; touches the vector ALU only when the tap / input tensor changes
    if (--left == 0) {
      if (cur_seg == 0 && cp0 < cpt) {
        cur_seg = 1;
      } else {
        cur_seg = 0;
        ++cur_tap;
      }
      if (cur_tap < p.n_taps) {
        a_setup(cur_tap, cur_seg);
        left = cur_seg ? cpt - cp0 : cp0;
        soff_a = 0;
      } else {
        left = 1 << 30;
      }
    }
  };

  const float pro_slope = p.pro_slope;
  float* const st_a = smem + lrow * ROW + (tid & 7) * 4;
  float* const st_b = st_a + BM * ROW;
  auto store = [&](Regs& R) {
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      float4 v = __builtin_bit_cast(float4, R.pa[i]);
      if constexpr (ACT == SRN_ACT_LEAKY) {  // 0 <= slope <= 1 (host-checked): max(x, slope x)
        v.x = fmaxf(v.x, v.x * pro_slope);
        v.y = fmaxf(v.y, v.y * pro_slope);
        v.z = fmaxf(v.z, v.z * pro_slope);
        v.w = fmaxf(v.w, v.w * pro_slope);
      }
      *reinterpret_cast<float4*>(st_a + i * 32 * ROW) = v;
    }
#pragma unroll
    for (int i = 0; i < C::B_LD; ++i) *reinterpret_cast<float4*>(st_b + i * 32 * ROW) = __builtin_bit_cast(float4, R.pb[i]);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) acc[m][n][r16] = 0.f;

  const int wq = wave % (C::WAVES_M * C::WAVES_N);  // wave inside its k group
  const int kg = wave / (C::WAVES_M * C::WAVES_N);  // k group
  const int wm0 = (wq / C::WAVES_N) * C::WM;
  const int wn0 = (wq % C::WAVES_N) * C::WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const float* const fa = smem + (wm0 + li) * ROW + 4 * lh + kg * (C::KK * 8);
  const float* const fb = smem + (BM + wn0 + li) * ROW + 4 * lh + kg * (C::KK * 8);

  auto compute = [&]() {
#pragma unroll
    for (int kk = 0; kk < C::KK; ++kk) {
      float4 af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(fa + m * 32 * ROW + kk * 8);
#pragma unroll
      for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float4*>(fb + n * 32 * ROW + kk * 8);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
    }
  };

  // One LDS stage, one register set: [write tile s | barrier | loads of tile s+1 issued, MFMAs of tile s | barrier].
  // Nothing overlaps inside the workgroup; the co-resident workgroups overlap each other (conv_fast.hip, NSTAGE = 1).
  if constexpr (C::PF == 2) {
    // two register sets: the loads of step s + 2 are issued inside step s.  For grids that leave a workgroup alone on
    // its CU (B = 1, split-K slices): nothing else covers a step's memory latency there
    Regs R0, R1;
    auto step = [&](Regs& R, const bool more) {
      store(R);
      __syncthreads();
      if (more) load_issue(R);
      compute();
      if (more) advance();
      __syncthreads();
    };
    if (n_steps == 1) {
      load_issue(R0);
      step(R0, false);
    } else if (n_steps > 1) {
      load_issue(R0);
      advance();
      load_issue(R1);
      advance();
      int s = 0;
      for (; s + 3 < n_steps; s += 2) {
        step(R0, true);
        step(R1, true);
      }
      if (n_steps - s == 3) {
        step(R0, true);
        step(R1, false);
        step(R0, false);
      } else {
        step(R0, false);
        step(R1, false);
      }
    }
  } else if (n_steps > 0) {
    Regs R;
    load_issue(R);
    advance();
    for (int s = 1; s < n_steps; ++s) {
      store(R);
      __syncthreads();
      load_issue(R);
      compute();
      advance();
      __syncthreads();
    }
    store(R);
    __syncthreads();
    compute();
  }

  if constexpr (C::KW > 1) {
    // join the k groups' partial accumulators in group order (bit-reproducible): group g > 0 parks its tile in LDS
    // as [register][lane] (conflict-free), group 0 adds and runs the epilogue
    __syncthreads();
    float* const park = smem + wq * (MT * NT * 16 * 64) + lane;
    if (kg == 1) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r16 = 0; r16 < 16; ++r16) park[((m * NT + n) * 16 + r16) * 64] = acc[m][n][r16];
    }
    __syncthreads();
    if (kg != 0) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) acc[m][n][r16] += park[((m * NT + n) * 16 + r16) * 64];
  }

  if (L.ksplit > 1) splitk_store<MT, NT>(p, acc, slice, z, t0, n0, wm0, wn0, lane);
  else f32_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

FDiv make_fdiv(const uint32_t d) {
  int lg = 0;
  while ((1u << lg) < d) ++lg;
  const int k = 26 + lg;
  return FDiv{(uint32_t)(((1ull << k) + d - 1) / d), (uint32_t)k};
}

template <class C, int ACT>
int launch_f32_2(const SrnConvParams& p, hipStream_t stream, const int ksplit) {
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_f32_kernel<C, ACT>), C::SMEM_BYTES)) return e;
  F32Launch L;
  L.m_tiles = (p.T_out + C::BM - 1) / C::BM;
  L.n_tiles = (p.N + C::BN - 1) / C::BN;
  L.ksplit = ksplit;
  const int cpt = p.C_in / BK;
  const int steps_all = p.n_taps * cpt;
  L.per_slice = (steps_all + ksplit - 1) / ksplit;
  const int64_t tiles_all = (int64_t)p.n_batch * p.n_head * L.m_tiles * L.n_tiles;
  const int64_t blocks = tiles_all * ksplit;
  if (blocks <= 0 || blocks >= (1ll << 26)) return 0;  // the magic divisions hold below 2^26: other kernels take it
  L.tiles_all = (int)tiles_all;
  L.d_per_z = make_fdiv((uint32_t)(L.m_tiles * L.n_tiles));
  L.d_band = make_fdiv((uint32_t)(TILE_BAND * L.n_tiles));
  L.d_head = make_fdiv((uint32_t)p.n_head);
  L.d_tiles_all = make_fdiv((uint32_t)tiles_all);
  L.d_cpt = make_fdiv((uint32_t)cpt);
  hipLaunchKernelGGL((conv_f32_kernel<C, ACT>), dim3((unsigned)blocks), dim3(256), C::SMEM_BYTES, stream, p, L);
  SRN_CHECK_LAUNCH();
  return 1;
}

template <class C>
int launch_f32(const SrnConvParams& p, hipStream_t stream, const int ksplit) {
  if (p.pro_act == SRN_ACT_NONE) return launch_f32_2<C, SRN_ACT_NONE>(p, stream, ksplit);
  return launch_f32_2<C, SRN_ACT_LEAKY>(p, stream, ksplit);
}

}  // namespace

// Returns 1 if the launch was handled, 0 if the shape is not eligible (the caller goes on to conv_fast.hip), < 0 on
// error.  `p` has been validated and defaulted by srn_conv_gemm.  Tile ids: 5 = 128 x 32, 6 = 128 x 128, 7 = 64 x 64, 9 = 64 x 128 (conv_fast.hip's
// single-stage ids, same results bit for bit), 10 = 32 x 64 with the step split over two wave pairs, 11 = 64 x 64 with
// two register sets (loads two steps ahead: small grids and split-K slices, where a workgroup is alone on its CU).
int srn_conv_f32_try(const SrnConvParams& p, int tile, hipStream_t stream, int ksplit) {
  if (p.precision != SRN_PREC_FP32 || p.w_nmajor) return 0;
  if (p.C_in % BK != 0 || p.C_in0 % BK != 0) return 0;
  if (p.C_w != p.C_in || p.ldw < p.n_taps * p.C_in) return 0;
  if (!(p.pro_act == SRN_ACT_NONE || (p.pro_act == SRN_ACT_LEAKY && p.pro_slope >= 0.f && p.pro_slope <= 1.f))) return 0;
  // 32-bit byte offsets inside one item of every operand
  const int64_t lim = 0x7fffffffll;
  if (((int64_t)p.T_in * p.ld_in0 + p.C_in) * 4 >= lim) return 0;
  if (p.in1 && ((int64_t)p.T_in * p.ld_in1 + p.C_in) * 4 >= lim) return 0;
  if (((int64_t)p.N * p.ldw + (int64_t)p.n_taps * p.C_in) * 4 >= lim) return 0;
  if (p.T_in >= (1 << 24) || p.N >= (1 << 24) || p.ld_in0 >= (1 << 22) || p.ld_in1 >= (1 << 22) || p.ldw >= (1 << 22)) return 0;
  const int64_t rows_all = (int64_t)(p.T_out - 1) * p.out_t_stride + p.out_t_off + 1;
  if ((rows_all * p.ld_out + p.N) * 4 >= lim || (rows_all * p.ld_res + p.N) * 4 >= lim ||
      (rows_all * p.ld_res2 + p.N) * 4 >= lim)
    return 0;
  switch (tile) {
    case 5: return p.geglu ? 0 : launch_f32<TCfg<128, 32, 32, 32, 1, 6>>(p, stream, ksplit);  // thin outputs (N = 32)
    case 6: return launch_f32<TCfg<128, 128, 64, 64, 1, 3>>(p, stream, ksplit);
    case 7: return launch_f32<TCfg<64, 64, 32, 32, 1, 6>>(p, stream, ksplit);
    case 9: return launch_f32<TCfg<64, 128, 32, 64, 1, 4>>(p, stream, ksplit);
    case 10: return p.geglu ? 0 : launch_f32<TCfg<32, 64, 32, 32, 2, 6>>(p, stream, ksplit);
    case 11: return launch_f32<TCfg<64, 64, 32, 32, 1, 5, 2>>(p, stream, ksplit);
    default: return 0;
  }
}
