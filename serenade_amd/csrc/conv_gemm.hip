// conv_gemm.hip — generalised implicit-GEMM conv1d / linear / batched GEMM for gfx950 (CDNA4),
// exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain).
//
// Layout: activations channels-last (z, t, c) so that a GEMM row (one frame) is one contiguous
// channel vector: the A tile of a conv tap is a run of consecutive, shifted frames — coalesced
// 128-B row segments, no im2col.  Weights are pre-packed [n][tap][c] (k contiguous) or, for the
// P·V product of attention, [k][n].
//
// Tiling: 256 threads = 4 wave64s per workgroup, BK = 32.  A/B tiles are staged through LDS with a
// 4-float row pad (144-B rows: ds_read_b128 of 16 distinct rows hits 16 distinct 4-bank slots), double
// buffered, with the next tile's global loads issued before the MFMA phase and written to LDS after it
// (issue-early / write-late).  Each lane reads 4 consecutive k per ds_read_b128; MFMA step s pairs
// k = 8*kk + s (lanes 0-31) with k = 8*kk + 4 + s (lanes 32-63) for A and B alike.
// The blockIdx -> tile map is XCD-aware (blocks that share an A tile land on one XCD's L2).
//
// Reference code this replaces: see include/serenade_hip.h (SrnConvParams).
#include "conv_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats)

template <int BM_, int BN_, int WM_, int WN_, bool NMAJ_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr bool NMAJ = NMAJ_;
  static constexpr int MT = WM / 32, NT = WN / 32;
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static constexpr int LDN = BN + 4;  // n-major B tile row (floats)
  static constexpr int A_STAGE = BM * LDK;
  static constexpr int B_STAGE = NMAJ ? BK * LDN : BN * LDK;
  static constexpr int SMEM_BYTES = 2 * (A_STAGE + B_STAGE) * (int)sizeof(float);
  static constexpr int A_LD = BM / 32;  // float4 loads per thread per stage
  static constexpr int B_LD = BN / 32;
};

__device__ __forceinline__ float4 act4(float4 v, int act, float slope) {
  if (act != SRN_ACT_NONE) {
    v.x = srn_act(v.x, act, slope);
    v.y = srn_act(v.y, act, slope);
    v.z = srn_act(v.z, act, slope);
    v.w = srn_act(v.w, act, slope);
  }
  return v;
}

// ACT: prologue activation compiled in: SRN_ACT_NONE, SRN_ACT_LEAKY, or -1 = decided at run time (SiLU / Mish:
// only the tiny time-embedding GEMMs use those, so only the small tile is instantiated with -1).
// PREC: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = split-bf16: every fp32 operand is staged in LDS as a
// (hi, lo) bf16 pair and each product is three v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi), fp32 accumulate:
// ~2^-17 relative error per product at 16/3 x the fp32 MFMA rate.
template <class C, int ACT, int PREC>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const SrnConvParams p, const int m_tiles,
                                                        const int n_tiles) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * C::A_STAGE;
  // split-bf16 stage: [A_hi | A_lo | B_hi | B_lo], 64 B per tile row
  constexpr int BF_STAGE = (BM + BN) * 128;
  unsigned char* smem_b = reinterpret_cast<unsigned char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int logical = xcd_logical_block();
  int z, mt_i, nt_i;
  tile_coords(logical, m_tiles, n_tiles, z, mt_i, nt_i);
  const int zb = z / p.n_head;
  const int zh = z - zb * p.n_head;
  const int t0 = mt_i * BM;
  const int n0 = nt_i * BN;

  const float* __restrict__ in0 = p.in0 + (int64_t)zb * p.in0_bs + (int64_t)zh * p.in0_hs;
  const float* __restrict__ in1 = p.in1 ? p.in1 + (int64_t)zb * p.in1_bs : nullptr;
  const float* __restrict__ wgt = p.w + (int64_t)zb * p.w_bs + (int64_t)zh * p.w_hs;
  const int T_in = p.T_in;
  int len_in = T_in;
  if (p.len_in) len_in = min(p.len_in[zb], T_in);

  const int n_chunks = (p.C_in + BK - 1) / BK;
  const int n_steps = p.n_taps * n_chunks;

  // ---- per-thread load coordinates
  const int c4 = tid & 7;    // float4 column inside the 32-wide k chunk
  const int lrow = tid >> 3;  // 0..31
  int a_tb[C::A_LD];          // input row before the tap offset, or INT_MIN/2 if the output row is invalid
#pragma unroll
  for (int i = 0; i < C::A_LD; ++i) {
    const int t = t0 + lrow + 32 * i;
    a_tb[i] = (t < p.T_out) ? t * p.in_stride : -(1 << 29);
  }

  // register staging set of the tile in flight
  struct Regs {
    float4 pa[C::A_LD];
    float4 pb[C::B_LD];
    unsigned a_ok, b_ok;  // validity bits of the tile rows staged in pa / pb
  };
  Regs R0;

  // Issue-only: every load targets an in-bounds (clamped) address and nothing here consumes a loaded value, so
  // no s_waitcnt is needed until store_step() -- the loads stay in flight under the MFMA phase.
  auto load_step = [&](int step, Regs& R) {
    const int tap = step / n_chunks;
    const int chunk = step - tap * n_chunks;
    const int ch = chunk * BK + c4 * 4;
    const int toff = p.tap_off[tap];
    {
      const float* src = in0;
      int ld = p.ld_in0;
      int c = ch;
      const bool cok = ch < p.C_in;
      if (cok && ch >= p.C_in0) {
        src = in1;
        ld = p.ld_in1;
        c = ch - p.C_in0;
      }
      if (!cok) c = 0;
      R.a_ok = 0;
#pragma unroll
      for (int i = 0; i < C::A_LD; ++i) {
        int ti = a_tb[i] + toff;
        if (p.pad_reflect) {  // 2: mirror at the item's own end (ragged batches), else at the tensor's end
          const int T_ref = p.pad_reflect == 2 ? len_in : T_in;
          if (ti < 0 && ti > -(1 << 28)) ti = -ti;
          if (ti >= T_ref) ti = 2 * (T_ref - 1) - ti;
        }
        const bool ok = cok && ti >= 0 && ti < len_in;
        R.a_ok |= (ok ? 1u : 0u) << i;
        ti = min(max(ti, 0), T_in - 1);
        R.pa[i] = *reinterpret_cast<const float4*>(src + (int64_t)ti * ld + c);
      }
    }
    if constexpr (!C::NMAJ) {
      const bool kok = ch < p.C_w;
      const int64_t kcol = (int64_t)tap * p.C_in + (kok ? ch : 0);
      R.b_ok = 0;
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) {
        int n = n0 + lrow + 32 * i;
        const bool ok = kok && n < p.N;
        R.b_ok |= (ok ? 1u : 0u) << i;
        n = min(n, p.N - 1);
        R.pb[i] = *reinterpret_cast<const float4*>(wgt + (int64_t)n * p.ldw + kcol);
      }
    } else {
      constexpr int F4_PER_ROW = BN / 4;
      R.b_ok = 0;
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) {
        const int f = tid + i * 256;
        const int krow = f / F4_PER_ROW;
        int n = n0 + (f % F4_PER_ROW) * 4;
        int k = chunk * BK + krow;
        const bool ok = k < p.C_w && n < p.N;
        R.b_ok |= (ok ? 1u : 0u) << i;
        k = min(k, p.C_w - 1);
        n = min(n, p.N - 4);
        R.pb[i] = *reinterpret_cast<const float4*>(wgt + (int64_t)k * p.ldw + n);
      }
    }
  };

  const int pro_act = p.pro_act;
  const float pro_slope = p.pro_slope;
  auto store_step = [&](int stage, Regs& R) {
    if constexpr (PREC == 1) {
      unsigned char* sa_hi = smem_b + stage * BF_STAGE;
      unsigned char* sa_lo = sa_hi + BM * 64;
      unsigned char* sb_hi = sa_lo + BM * 64;
      unsigned char* sb_lo = sb_hi + BN * 64;
#pragma unroll
      for (int i = 0; i < C::A_LD; ++i) {
        float4 v = sel4((R.a_ok >> i) & 1u, R.pa[i]);
        if constexpr (ACT == SRN_ACT_LEAKY) {
          v.x = v.x > 0.f ? v.x : v.x * pro_slope;
          v.y = v.y > 0.f ? v.y : v.y * pro_slope;
          v.z = v.z > 0.f ? v.z : v.z * pro_slope;
          v.w = v.w > 0.f ? v.w : v.w * pro_slope;
        } else if constexpr (ACT < 0) {
          v = act4(v, pro_act, pro_slope);
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        const int off = bf_off(lrow + 32 * i, c4 * 4);
        *reinterpret_cast<bf16x4*>(sa_hi + off) = hi;
        *reinterpret_cast<bf16x4*>(sa_lo + off) = lo;
      }
      if constexpr (!C::NMAJ) {
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
          bf16x4 hi, lo;
          split4(sel4((R.b_ok >> i) & 1u, R.pb[i]), hi, lo);
          const int off = bf_off(lrow + 32 * i, c4 * 4);
          *reinterpret_cast<bf16x4*>(sb_hi + off) = hi;
          *reinterpret_cast<bf16x4*>(sb_lo + off) = lo;
        }
      } else {
        // n-major source (P.V): transpose while staging -> the same [n][k] image
        constexpr int F4_PER_ROW = BN / 4;
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
          const int f = tid + i * 256;
          const int krow = f / F4_PER_ROW;
          const int nn = (f % F4_PER_ROW) * 4;
          bf16x4 hi, lo;
          split4(sel4((R.b_ok >> i) & 1u, R.pb[i]), hi, lo);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int off = bf_off(nn + e, krow);
            *reinterpret_cast<__bf16*>(sb_hi + off) = hi[e];
            *reinterpret_cast<__bf16*>(sb_lo + off) = lo[e];
          }
        }
      }
      return;
    }
    float* a = As + stage * C::A_STAGE;
    float* b = Bs + stage * C::B_STAGE;
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      float4 v = sel4((R.a_ok >> i) & 1u, R.pa[i]);
      // act(0) == 0 for every supported activation, so masked rows stay zero
      if constexpr (ACT == SRN_ACT_LEAKY) {
        v.x = v.x > 0.f ? v.x : v.x * pro_slope;
        v.y = v.y > 0.f ? v.y : v.y * pro_slope;
        v.z = v.z > 0.f ? v.z : v.z * pro_slope;
        v.w = v.w > 0.f ? v.w : v.w * pro_slope;
      } else if constexpr (ACT < 0) {
        v = act4(v, pro_act, pro_slope);
      }
      *reinterpret_cast<float4*>(a + (lrow + 32 * i) * LDK + c4 * 4) = v;
    }
    if constexpr (!C::NMAJ) {
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i)
        *reinterpret_cast<float4*>(b + (lrow + 32 * i) * LDK + c4 * 4) = sel4((R.b_ok >> i) & 1u, R.pb[i]);
    } else {
      constexpr int F4_PER_ROW = BN / 4;
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) {
        const int f = tid + i * 256;
        *reinterpret_cast<float4*>(b + (f / F4_PER_ROW) * C::LDN + (f % F4_PER_ROW) * 4) =
            sel4((R.b_ok >> i) & 1u, R.pb[i]);
      }
    }
  };

  // ---- accumulators
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int wm0 = (wave / C::WAVES_N) * C::WM;
  const int wn0 = (wave % C::WAVES_N) * C::WN;
  const int li = lane & 31;
  const int lh = lane >> 5;

  auto compute = [&](const int cur) {
    if constexpr (PREC == 1) {
      const unsigned char* sa_hi = smem_b + cur * BF_STAGE;
      const unsigned char* sa_lo = sa_hi + BM * 64;
      const unsigned char* sb_hi = sa_lo + BM * 64;
      const unsigned char* sb_lo = sb_hi + BN * 64;
      const int sw = (li >> 2) & 3;  // row swizzle key (tile row offsets are multiples of 32)
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const int choff = (((kk * 2 + lh) ^ sw) & 3) << 4;
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int o = (wm0 + m * 32 + li) * 64 + choff;
          ah[m] = *reinterpret_cast<const bf16x8*>(sa_hi + o);
          al[m] = *reinterpret_cast<const bf16x8*>(sa_lo + o);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int o = (wn0 + n * 32 + li) * 64 + choff;
          bh[n] = *reinterpret_cast<const bf16x8*>(sb_hi + o);
          bl[n] = *reinterpret_cast<const bf16x8*>(sb_lo + o);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
      }
    } else {
    const float* a = As + cur * C::A_STAGE + (wm0 + li) * LDK + 4 * lh;
    const float* b = C::NMAJ ? Bs + cur * C::B_STAGE + (4 * lh) * C::LDN + wn0 + li
                             : Bs + cur * C::B_STAGE + (wn0 + li) * LDK + 4 * lh;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a + m * 32 * LDK + kk * 8);
      if constexpr (!C::NMAJ) {
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float4*>(b + n * 32 * LDK + kk * 8);
      } else {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float* bp = b + (kk * 8) * C::LDN + n * 32;
          bf[n] = make_float4(bp[0], bp[C::LDN], bp[2 * C::LDN], bp[3 * C::LDN]);
        }
      }
      // k-step outermost: consecutive MFMAs hit different accumulators (no back-to-back dependent issue)
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
    }
  };

  // Pipeline: LDS double-buffered; while tile s is multiplied out of LDS[s & 1], tile s+1 is in flight from
  // global memory into registers and is staged (activation / split / ds_write) after the MFMA phase.
  // (A second register set = prefetch distance 2 was measured: +3 % only, and the 2x-unrolled loop made hipcc
  // shuffle the 64 accumulator registers every iteration.)
  if constexpr (PREC == 1) {
    // split-bf16: the MFMA phase is short (24 x 32 cycles), so the staging VALU work (split + ds_write) of the
    // NEXT tile must overlap it inside the same wave: tile s+1 already sits in one register set (its loads were
    // issued a whole iteration earlier) while tile s+2 is being fetched into the other.
    Regs R1;
    // MFMA phase of tile s and staging of tile s+1 in ONE basic block, interleaved by the scheduler:
    // per MFMA (32 cycles, 8 of which block vector issue) ~7 VALU + LDS traffic ride along.
    auto fused = [&](const int cs, const int ss, Regs& R) {
      compute(cs);
      store_step(ss, R);
      constexpr int N_MFMA = MT * NT * 6;
      constexpr int N_DSR = (MT + NT) * 4;
      constexpr int VPM = ((C::A_LD + C::B_LD) * 20 + N_MFMA - 1) / N_MFMA;
      __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT) * 2, 0);  // fragments of the first k16 step
#pragma unroll
      for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < N_DSR - (MT + NT) * 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    };
    load_step(0, R0);
    store_step(0, R0);
    if (n_steps > 1) load_step(1, R0);
    __syncthreads();
    int step = 0;
    // invariant: LDS[0] holds tile `step`; R0 holds the raw tile step+1 (if it exists)
    for (; step + 2 < n_steps; step += 2) {
      load_step(step + 2, R1);
      fused(0, 1, R0);
      __syncthreads();
      if (step + 3 < n_steps) load_step(step + 3, R0);
      fused(1, 0, R1);
      __syncthreads();
    }
    compute(0);
    if (step + 1 < n_steps) {
      store_step(1, R0);
      __syncthreads();
      compute(1);
    }
  } else {
    load_step(0, R0);
    store_step(0, R0);
    __syncthreads();
    for (int step = 0; step < n_steps; ++step) {
      const int cur = step & 1;
      if (step + 1 < n_steps) load_step(step + 1, R0);
      compute(cur);
      if (step + 1 < n_steps) store_step(cur ^ 1, R0);
      __syncthreads();
    }
  }

  conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

struct TileInfo {
  int id, bm, bn, wn;
  float base[3];  // measured relative efficiency of the tile shape on large grids: [fp32, split-bf16, bf16x6]
};
// id: 1 128x128 (64x64/wave) | 2 128x64 (32x64) | 3 64x128 (32x64) | 4 64x64 (32x32) | 5 128x32 (32x32)
// (priors from tools/opbench.py --sweep on MI355X: in fp32 the MFMA phase is long and the small tile loses nothing;
//  in split-bf16 the kernel is L2-traffic sensitive and bigger tiles win)
const TileInfo kTiles[] = {{1, 128, 128, 64, {1.00f, 1.00f, 1.00f}}, {2, 128, 64, 64, {0.90f, 0.95f, 0.95f}},
                           {3, 64, 128, 64, {0.92f, 0.97f, 0.97f}},  {4, 64, 64, 32, {0.97f, 0.85f, 0.90f}},
                           {5, 128, 32, 32, {0.85f, 0.75f, 0.80f}}};

template <class C, int ACT, int PREC>
int launch_prec(const SrnConvParams& p, hipStream_t stream) {
  constexpr int SMEM = PREC == 1 ? 2 * (C::BM + C::BN) * 128 : C::SMEM_BYTES;
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_gemm_kernel<C, ACT, PREC>), SMEM)) return e;
  const int m_tiles = (p.T_out + C::BM - 1) / C::BM;
  const int n_tiles = (p.N + C::BN - 1) / C::BN;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m_tiles * n_tiles;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_gemm: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL((conv_gemm_kernel<C, ACT, PREC>), dim3((unsigned)blocks), dim3(256), SMEM, stream, p, m_tiles,
                     n_tiles);
  SRN_CHECK_LAUNCH();
  return 0;
}

template <class C, int ACT>
int launch_act(const SrnConvParams& p, hipStream_t stream) {
  if (p.precision == SRN_PREC_BF16X3) return launch_prec<C, ACT, 1>(p, stream);
  return launch_prec<C, ACT, 0>(p, stream);
}

template <class C>
int launch(const SrnConvParams& p, hipStream_t stream) {
  if (p.pro_act == SRN_ACT_NONE) return launch_act<C, SRN_ACT_NONE>(p, stream);
  if (p.pro_act == SRN_ACT_LEAKY) return launch_act<C, SRN_ACT_LEAKY>(p, stream);
  // SiLU / Mish prologue: run-time activation, small tile only
  return launch_act<Cfg<64, 64, 32, 32, C::NMAJ>, -1>(p, stream);
}

int pick_tile(const SrnConvParams& p) {
  // Exact fp32: the single-LDS-stage forms (ids 6-9, conv_fast.hip) win on every shape of the path (opbench --fp32
  // --sweep, r3): an fp32 MFMA is 64 cycles, so the loop is MFMA-bound with one stage, and the halved LDS footprint lets
  // 4-6 workgroups share a CU -- their staggered fills / epilogues cover each other, which two double-buffered
  // workgroups running in lockstep do not.  64 x 64 (id 7) for N = 512 .. 6144 (+4 .. +9 % over its two-stage form),
  // 64 x 128 (id 9) for the GEGLU projection and the N = 128 / 256 HiFi-GAN stages (+3 .. +6 %).
  // They need company: with fewer than ~3 workgroups per CU (B = 1, short utterances) nothing covers a single-stage
  // workgroup's fill and the double-buffered forms below are ahead (B = 1 x T = 256: 16.7 vs 18.3 ms).
  if (p.precision == SRN_PREC_FP32 && !p.w_nmajor && p.C_in % 32 == 0) {
    const int64_t rows = (int64_t)p.n_batch * p.n_head * ((p.T_out + 63) / 64);
    const int64_t blocks9 = rows * ((p.N + 127) / 128), blocks7 = rows * ((p.N + 63) / 64);
    if (p.geglu && blocks9 >= 512) return 9;
    if (p.N % 128 == 0 && p.N <= 256 && blocks9 >= 512) return 9;
    if (!p.geglu && p.N % 64 == 0 && blocks7 >= 512) {
      // every tile of the launch is co-resident (<= 6 per CU): the launch lasts as long as the fullest CU.  640 tiles
      // of 64 x 64 (5120 rows x 512 columns: the half-resolution levels) are 3 on some CUs and 2 on others; the same
      // output as 32 x 64 tiles (id 10, conv_f32.hip) is 5 on every CU
      const int64_t blocks10 = 2 * blocks7;
      if (blocks10 <= 6 * 256 && p.T_out % 32 == 0 &&
          1.05 * 0.5 * (double)((blocks10 + 255) / 256) < (double)((blocks7 + 255) / 256))
        return 10;
      return 7;
    }
  }
  float best = -1.f;
  int best_id = 4;
  const double z = (double)p.n_batch * p.n_head;
  for (const TileInfo& t : kTiles) {
    if (p.geglu && t.wn < 64) continue;
    if (p.w_nmajor && !(t.id == 1 || t.id == 3 || t.id == 4)) continue;
    const double mt = (p.T_out + t.bm - 1) / t.bm, nt = (p.N + t.bn - 1) / t.bn;
    const double blocks = z * mt * nt;
    const double useful = ((double)p.T_out * p.N) / (mt * t.bm * nt * t.bn);
    const double rounds = (blocks + 255.0) / 256.0;
    double quant = blocks / (256.0 * (double)(int64_t)rounds);
    // fewer blocks than CUs idles CUs outright; beyond one round, co-resident blocks absorb part of the tail
    if (blocks > 256.0) quant = 0.35 + 0.65 * quant;
    float score = (float)(useful * quant) * t.base[p.precision == SRN_PREC_BF16X3 ? 1 : (p.precision == SRN_PREC_BF16X6 ? 2 : 0)];
    // both operands split in the loop (Q K^T, P V): the 64x128 tile measured 5-10 % ahead of 128x128
    if (p.precision == SRN_PREC_BF16X3 && (p.w_hi == nullptr || p.w_bs != 0 || p.w_hs != 0) && t.id == 1)
      score *= 0.92f;
    if (score > best) {
      best = score;
      best_id = t.id;
    }
  }
  // bf16x6: wherever the model settles on the 64 x 64 tile, its single-stage form (id 7) is ahead on every shape of the
  // path (+1 .. +14 %, opbench --bf16x6 --sweep): same reason as in exact fp32, six MFMAs per product keep the loop
  // MFMA-bound.  (Split-bf16 x3 is LDS- / power-bound instead and the single-stage forms lose as often as they win.)
  if (best_id == 4 && p.precision == SRN_PREC_BF16X6 && !p.w_nmajor && p.C_in % 32 == 0 &&
      (int64_t)p.n_batch * p.n_head * ((p.T_out + 63) / 64) * ((p.N + 63) / 64) >= 512)
    best_id = 7;
  return best_id;
}

}  // namespace

extern "C" int srn_conv_gemm(const SrnConvParams* pp, void* stream_) {
  SRN_CHECK_ARG(pp != nullptr, "conv_gemm: null params");
  SrnConvParams p = *pp;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  SRN_CHECK_ARG(p.in0 && p.w && p.out, "conv_gemm: null in0/w/out");
  SRN_CHECK_ARG(p.n_batch > 0 && p.n_head > 0 && p.T_in > 0 && p.T_out > 0 && p.N > 0, "conv_gemm: bad sizes");
  SRN_CHECK_ARG(p.C_in > 0 && p.C_in % 4 == 0, "conv_gemm: C_in (%d) must be a positive multiple of 4", p.C_in);
  SRN_CHECK_ARG(p.n_taps >= 1 && p.n_taps <= SRN_MAX_TAPS, "conv_gemm: n_taps %d out of range", p.n_taps);
  SRN_CHECK_ARG(p.ld_in0 % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in0) & 15) == 0 && p.in0_bs % 4 == 0 &&
                    p.in0_hs % 4 == 0,
                "conv_gemm: in0 must be 16-byte aligned with ld %% 4 == 0");
  if (p.C_in0 <= 0 || p.C_in0 > p.C_in) p.C_in0 = p.C_in;
  if (p.C_in0 < p.C_in) {
    SRN_CHECK_ARG(p.in1 != nullptr && p.C_in0 % 32 == 0 && p.ld_in1 % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(p.in1) & 15) == 0 && p.in1_bs % 4 == 0,
                  "conv_gemm: concat input needs in1, C_in0 %% 32 == 0 and aligned in1");
  }
  if (p.C_w <= 0 || p.C_w > p.C_in) p.C_w = p.C_in;
  SRN_CHECK_ARG(p.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15) == 0 && p.w_bs % 4 == 0 && p.w_hs % 4 == 0,
                "conv_gemm: w must be 16-byte aligned with ldw %% 4 == 0");
  if (p.w_nmajor) {
    SRN_CHECK_ARG(p.n_taps == 1 && p.N % 4 == 0, "conv_gemm: n-major weights need n_taps == 1 and N %% 4 == 0");
  } else {
    SRN_CHECK_ARG(p.C_w % 4 == 0, "conv_gemm: C_w must be a multiple of 4 for k-major weights");
  }
  if (p.in_stride <= 0) p.in_stride = 1;
  if (p.out_t_stride <= 0) p.out_t_stride = 1;
  if (p.N_out <= 0) p.N_out = p.geglu ? p.N / 2 : p.N;
  if (p.geglu)
    SRN_CHECK_ARG(p.N % 64 == 0 && p.res_mode == SRN_RES_NONE && p.res2 == nullptr && p.post == SRN_POST_NONE,
                  "conv_gemm: GEGLU needs N %% 64 == 0 and no residual / post op");
  if (p.res_mode != SRN_RES_NONE) SRN_CHECK_ARG(p.res != nullptr, "conv_gemm: res_mode set but res is null");
  if (p.out_tr)
    SRN_CHECK_ARG(!p.geglu && p.res_mode == SRN_RES_NONE && p.res2 == nullptr && p.post == SRN_POST_NONE &&
                      p.out_t_stride == 1 && p.out_t_off == 0 && p.n_head == 1 && p.gn_partials == nullptr &&
                      p.out_tr_col0 >= 0 && p.out_tr_col0 % 32 == 0 && p.ld_out_tr % 4 == 0 && p.ld_out_tr >= p.T_out &&
                      (reinterpret_cast<uintptr_t>(p.out_tr) & 15) == 0 && p.out_tr_bs % 4 == 0,
                  "conv_gemm: out_tr needs the plain epilogue, col0 %% 32 == 0 and a 16-byte aligned V^T with ld %% 4 == 0");
  if (p.gn_partials) SRN_CHECK_ARG(p.N % 32 == 0 && !p.geglu, "conv_gemm: gn_partials needs N %% 32 == 0");
  if (p.pad_reflect) {
    for (int i = 0; i < p.n_taps; ++i)
      SRN_CHECK_ARG(p.tap_off[i] > -p.T_in && p.tap_off[i] < p.T_in, "conv_gemm: reflect pad wider than the input");
  }

  int tile = p.tile > 0 ? p.tile : pick_tile(p);
  if (p.geglu && !(tile == 1 || tile == 2 || tile == 3 || tile == 6 || tile == 8 || tile == 9)) tile = 1;
  if (p.ws != nullptr && p.tile <= 0 && p.no_halo != 3) {
    // small grids with a deep contraction (B = 1 / short utterances): slice K over extra workgroups, reduce after
    const int ks = srn_splitk_plan(p);
    if (ks > 1 && p.ws_bytes >= srn_splitk_bytes(p, ks)) {
      // exact fp32: conv_f32.hip's 64 x 64 tile with loads two steps ahead (id 11; B = 1 x T = 256: 16.8 -> 15.8 ms
      // against conv_fast.hip's double-buffered tile here and for the unsplit small grids below)
      int r = 0;
      if (p.precision == SRN_PREC_FP32 && p.no_halo != 5) r = srn_conv_f32_try(p, 11, stream, ks);
      if (r == 0) r = srn_conv_fast_try(p, 4, stream, ks);
      if (r < 0) return r;
      if (r == 1) return srn_splitk_reduce(p, ks, stream);
    }
  }
  if (p.no_halo != 1 && p.no_halo != 3) {
    // thin convs (<= 64 channels in and out): persistent strip kernel with LDS-resident weights
    const int r = srn_conv_strip_try(p, stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  if (p.no_halo != 1 && p.no_halo != 3) {
    // stride-1 multi-tap convs in split-bf16: stage the receptive-field tile once per channel chunk
    const int r = srn_conv_halo_try(p, tile, stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  if (p.no_halo != 3) {  // 3: generic kernel only (testing / A-B timing)
    if (p.precision == SRN_PREC_FP32 && p.no_halo != 5) {  // 5: conv_fast.hip's fp32 form (A-B timing against conv_f32.hip)
      const int r = srn_conv_f32_try(p, tile == 4 ? 11 : tile, stream, 1);  // 4: small grids -> its two-steps-ahead form
      if (r != 0) return r < 0 ? r : 0;
    }
    if (tile == 10) tile = 7;  // the split-step tile exists only in conv_f32.hip
    if (tile == 11) tile = 4;  // ... and so does the two-steps-ahead tile: conv_fast.hip's double-buffered twin
    const int r = srn_conv_fast_try(p, tile, stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  // single-stage ids exist only in conv_fast.hip: their two-stage twins here
  tile = tile == 6 ? 1 : (tile == 7 || tile == 10 || tile == 11) ? 4 : tile == 8 ? 2 : tile == 9 ? 3 : tile;
  if (p.w_nmajor) {
    switch (tile) {
      case 1: return launch<Cfg<128, 128, 64, 64, true>>(p, stream);
      case 3: return launch<Cfg<64, 128, 32, 64, true>>(p, stream);
      default: return launch<Cfg<64, 64, 32, 32, true>>(p, stream);
    }
  }
  switch (tile) {
    case 1: return launch<Cfg<128, 128, 64, 64, false>>(p, stream);
    case 2: return launch<Cfg<128, 64, 32, 64, false>>(p, stream);
    case 3: return launch<Cfg<64, 128, 32, 64, false>>(p, stream);
    case 4: return launch<Cfg<64, 64, 32, 32, false>>(p, stream);
    case 5: return launch<Cfg<128, 32, 32, 32, false>>(p, stream);
    default: break;
  }
  srn_set_error("conv_gemm: unknown tile id %d", tile);
  return -1;
}

extern "C" int64_t srn_conv_gemm_workspace_bytes(const SrnConvParams* pp) {
  if (pp == nullptr) return 0;
  SrnConvParams p = *pp;
  if (p.C_in0 <= 0 || p.C_in0 > p.C_in) p.C_in0 = p.C_in;
  if (p.C_w <= 0 || p.C_w > p.C_in) p.C_w = p.C_in;
  if (p.n_head <= 0 || p.n_batch <= 0 || p.T_out <= 0 || p.N <= 0 || p.C_in <= 0 || p.n_taps <= 0) return 0;
  const int ks = srn_splitk_plan(p);
  return ks > 1 ? srn_splitk_bytes(p, ks) : 0;
}
