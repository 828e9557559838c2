// conv_ws.hip -- wave-specialised form of conv_fast.hip's split-bf16 contraction (bf16x3 and bf16x6) for the large
// regular GEMMs of the path (C_in a multiple of 32, k-major weights, 128 x 128 tiles).
//
// Why: conv_fast's loop is ISSUE-bound -- every wave stages (global loads, fp32 -> bf16 split, LDS writes: ~150
// instructions) AND multiplies (24 MFMAs) per 32-deep step, and the PMC profile of the round-2 build shows the MFMA
// pipe busy 35 % with 40 % of wave time parked in waits (profiles/r2_a_pmc_sq_bf16x3.txt).  Here a workgroup is
// 512 threads: waves 0-3 are CONSUMERS (one 64 x 64 accumulator tile each; their instruction stream is LDS fragment
// reads + MFMAs only), waves 4-7 are PRODUCERS (operand cursors, global loads two steps ahead into two register sets,
// split, LDS writes; no accumulators).  The hardware places the two halves of a workgroup on the same four SIMDs, so
// every SIMD holds MFMA-only and VALU-only waves side by side -- the pairing the CU runs concurrently.  One barrier
// per step, LDS double-buffered exactly like conv_fast (stage s is multiplied while stage s+1 is written).
// The LDS image, the cursor arithmetic, the MFMA order and the epilogue are conv_fast's, so results are bit-identical
// to conv_fast's 128 x 128 tile.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int BK = 32;
constexpr int BM = 128, BN = 128, MT = 2, NT = 2;  // consumer wave tile 64 x 64
constexpr int A_LD = BM / 32, B_LD = BN / 32;      // 16-B loads per producer thread per step

__device__ __attribute__((aligned(256))) float g_zero_ws[64];

typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2w(const float a, const float b, unsigned& hi, unsigned& lo) {
  const f32x2w v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2w));
  f32x2w hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2w l = v - hf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2w));
}

__device__ __forceinline__ void split3w(const float a, const float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2w v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2w));
  f32x2w hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2w r1 = v - hf;
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2w));
  f32x2w mf;
  mf.x = __builtin_bit_cast(float, mid << 16);
  mf.y = __builtin_bit_cast(float, mid & 0xffff0000u);
  const f32x2w r2 = r1 - mf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2w));
}

// NPL: bf16 planes per operand (2 = bf16x3, 3 = bf16x6).  ACT / WPL as in conv_fast.hip.
template <int NPL, int ACT, bool WPL>
__global__ __launch_bounds__(512, NPL == 2 ? 4 : 2) void conv_ws_kernel(const SrnConvParams p, const int m_tiles,
                                                                        const int n_tiles) {
  constexpr int STAGE = (BM + BN) * 64 * NPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int logical = xcd_logical_block();
  int z, mt_i, nt_i;
  tile_coords(logical, m_tiles, n_tiles, z, mt_i, nt_i);
  const int zb = z / p.n_head;
  const int zh = z - zb * p.n_head;
  const int t0 = mt_i * BM;
  const int n0 = nt_i * BN;
  const int cpt = p.C_in / BK;
  const int cp0 = p.C_in0 / BK;
  const int n_steps = p.n_taps * cpt;

  if (producer) {
    // =============================================================== producers: cursors, loads, split, LDS writes
    const int ptid = tid & 255;
    const float* in0 = p.in0 + (int64_t)zb * p.in0_bs + (int64_t)zh * p.in0_hs;
    const float* in1 = p.in1 ? p.in1 + (int64_t)zb * p.in1_bs : nullptr;
    const int T_in = p.T_in;
    int len_in = T_in;
    if (p.len_in) len_in = min(p.len_in[zb], T_in);
    const int c4 = ptid & 7;
    const int lrow = ptid >> 3;
    int a_tb[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) a_tb[i] = min(t0 + lrow + 32 * i, p.T_out - 1) * p.in_stride;
    const float* aptr[A_LD];
    int abump[A_LD];
    int cur_tap = 0, cur_seg = 0, left = 0;
    auto a_setup = [&](const int tap, const int seg) {
      const float* src = seg == 0 ? in0 : in1;
      const int ld = seg == 0 ? p.ld_in0 : p.ld_in1;
      const int toff = p.tap_off[tap];
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int ti = a_tb[i] + toff;
        if (p.pad_reflect) {
          if (ti < 0) ti = -ti;
          if (ti >= T_in) ti = 2 * (T_in - 1) - ti;
        }
        const bool ok = ti >= 0 && ti < len_in;
        aptr[i] = ok ? src + (int64_t)ti * ld + c4 * 4 : g_zero_ws + c4 * 4;
        abump[i] = ok ? BK : 0;
      }
      left = seg == 0 ? cp0 : cpt - cp0;
    };
    a_setup(0, 0);
    const float* bptr[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int n = min(n0 + lrow + 32 * i, p.N - 1);
      if constexpr (WPL)
        bptr[i] = reinterpret_cast<const float*>(p.w_hi) + ((int64_t)n * n_steps * 32 + c4 * 4);
      else
        bptr[i] = p.w + (int64_t)zb * p.w_bs + (int64_t)zh * p.w_hs + (int64_t)n * p.ldw + c4 * 4;
    }
    int bbump = BK;
    constexpr int B_LD2 = (WPL && NPL == 3) ? BN / 64 : 0;  // bf16x6: lo weight plane, 64-B lines (see conv_fast.hip)
    const float* cptr[B_LD2 > 0 ? B_LD2 : 1];
    const int c_q4 = ptid & 3, c_row = ptid >> 2;
    if constexpr (B_LD2 > 0) {
#pragma unroll
      for (int j = 0; j < B_LD2; ++j) {
        const int n = min(n0 + c_row + 64 * j, p.N - 1);
        cptr[j] = reinterpret_cast<const float*>(p.w_lo) + ((int64_t)n * n_steps) * 16 + c_q4 * 4;
      }
    }
    struct Regs {
      float4 pa[A_LD];
      float4 pb[B_LD];
      float4 pc[B_LD2 > 0 ? B_LD2 : 1];
    };
    auto load_issue = [&](Regs& R) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        R.pa[i] = *reinterpret_cast<const float4*>(aptr[i]);
        aptr[i] += abump[i];
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        R.pb[i] = *reinterpret_cast<const float4*>(bptr[i]);
        bptr[i] += bbump;
      }
      if constexpr (B_LD2 > 0) {
#pragma unroll
        for (int j = 0; j < B_LD2; ++j) {
          R.pc[j] = *reinterpret_cast<const float4*>(cptr[j]);
          cptr[j] += bbump >> 1;
        }
      }
    };
    auto cursor_advance = [&]() {
      if (--left == 0) {
        if (cur_seg == 0 && cp0 < cpt) {
          cur_seg = 1;
        } else {
          cur_seg = 0;
          ++cur_tap;
        }
        if (cur_tap < p.n_taps) {
          a_setup(cur_tap, cur_seg);
        } else {  // past the end: park on valid memory, keep the loads unconditional (counted vmcnt)
#pragma unroll
          for (int i = 0; i < A_LD; ++i) {
            aptr[i] = g_zero_ws + c4 * 4;
            abump[i] = 0;
          }
#pragma unroll
          for (int i = 0; i < B_LD; ++i) bptr[i] -= BK;
          if constexpr (B_LD2 > 0) {
#pragma unroll
            for (int j = 0; j < B_LD2; ++j) cptr[j] -= BK / 2;
          }
          bbump = 0;
          left = 1 << 30;
        }
      }
    };
    auto load = [&](Regs& R) {
      load_issue(R);
      cursor_advance();
    };
    const int pro_act = p.pro_act;
    const float pro_slope = p.pro_slope;
    const int st_off = bf_off(lrow, c4 * 4);
    const int stb_off = bf_off(lrow, (c4 & 3) * 8);
    const int stc_off = bf_off(c_row, c_q4 * 8);
    auto store = [&](const int stage, Regs& R) {
      unsigned char* sa_hi = smem_w + stage * STAGE;
      unsigned char* sa_lo = sa_hi + (NPL - 1) * BM * 64;
      unsigned char* sb_hi = sa_hi + NPL * BM * 64;
      unsigned char* sb_lo = sb_hi + (NPL - 1) * BN * 64;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        float4 v = R.pa[i];
        if constexpr (ACT == SRN_ACT_LEAKY) {
          v = leaky4(v, pro_slope);
        } else if constexpr (ACT < 0) {
          v.x = srn_act(v.x, pro_act, pro_slope);
          v.y = srn_act(v.y, pro_act, pro_slope);
          v.z = srn_act(v.z, pro_act, pro_slope);
          v.w = srn_act(v.w, pro_act, pro_slope);
        }
        if constexpr (NPL == 3) {
          uint2 hi, mid, lo;
          split3w(v.x, v.y, hi.x, mid.x, lo.x);
          split3w(v.z, v.w, hi.y, mid.y, lo.y);
          *reinterpret_cast<uint2*>(sa_hi + st_off + i * 2048) = hi;
          *reinterpret_cast<uint2*>(sa_hi + BM * 64 + st_off + i * 2048) = mid;
          *reinterpret_cast<uint2*>(sa_lo + st_off + i * 2048) = lo;
        } else {
          uint2 hi, lo;
          split2w(v.x, v.y, hi.x, lo.x);
          split2w(v.z, v.w, hi.y, lo.y);
          *reinterpret_cast<uint2*>(sa_hi + st_off + i * 2048) = hi;
          *reinterpret_cast<uint2*>(sa_lo + st_off + i * 2048) = lo;
        }
      }
      if constexpr (WPL) {
        unsigned char* dst = (c4 < 4 ? sb_hi : sb_hi + BN * 64) + stb_off;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
          float4 v = R.pb[i];
          asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));  // keep the staged set out of scratch
          *reinterpret_cast<float4*>(dst + i * 2048) = v;
        }
        if constexpr (B_LD2 > 0) {
#pragma unroll
          for (int j = 0; j < B_LD2; ++j) {
            float4 v = R.pc[j];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            *reinterpret_cast<float4*>(sb_lo + stc_off + j * 4096) = v;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
          const float4 v = R.pb[i];
          if constexpr (NPL == 3) {
            uint2 hi, mid, lo;
            split3w(v.x, v.y, hi.x, mid.x, lo.x);
            split3w(v.z, v.w, hi.y, mid.y, lo.y);
            *reinterpret_cast<uint2*>(sb_hi + st_off + i * 2048) = hi;
            *reinterpret_cast<uint2*>(sb_hi + BN * 64 + st_off + i * 2048) = mid;
            *reinterpret_cast<uint2*>(sb_lo + st_off + i * 2048) = lo;
          } else {
            uint2 hi, lo;
            split2w(v.x, v.y, hi.x, lo.x);
            split2w(v.z, v.w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(sb_hi + st_off + i * 2048) = hi;
            *reinterpret_cast<uint2*>(sb_lo + st_off + i * 2048) = lo;
          }
        }
      }
    };

    // register set k & 1 carries step k: steps s + 2 and s + 3 are in flight while step s + 1 is written
    Regs R0, R1;
    load(R0);       // step 0
    store(0, R0);
    load(R1);       // step 1
    load(R0);       // step 2
    __syncthreads();  // stage 0 visible
    int step = 0;
    for (; step + 2 <= n_steps; step += 2) {
      store(1, R1);  // step + 1 (a parked dummy past the end: written, never multiplied)
      load(R1);      // step + 3
      __syncthreads();
      store(0, R0);  // step + 2
      load(R0);      // step + 4
      __syncthreads();
    }
    if (step < n_steps) __syncthreads();  // odd tail: the consumers' last step
    return;
  }

  // ================================================================= consumers: LDS fragments -> MFMA, epilogue
  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const int wm0 = (wave >> 1) * 64;
  const int wn0 = (wave & 1) * 64;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int sw = (li >> 2) & 3;
  const int fr_a = (wm0 + li) * 64;
  const int fr_b = (wn0 + li) * 64;
  auto compute = [&](const int stage) {
    const unsigned char* sa_hi = smem_w + stage * STAGE;
    const unsigned char* sa_lo = sa_hi + (NPL - 1) * BM * 64;
    const unsigned char* sb_hi = sa_hi + NPL * BM * 64;
    const unsigned char* sb_lo = sb_hi + (NPL - 1) * BN * 64;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int choff = (((kk * 2 + lh) ^ sw) & 3) << 4;
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] = *reinterpret_cast<const bf16x8*>(sa_hi + fr_a + m * 2048 + choff);
        al[m] = *reinterpret_cast<const bf16x8*>(sa_lo + fr_a + m * 2048 + choff);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bh[n] = *reinterpret_cast<const bf16x8*>(sb_hi + fr_b + n * 2048 + choff);
        bl[n] = *reinterpret_cast<const bf16x8*>(sb_lo + fr_b + n * 2048 + choff);
      }
#define SRN_WS_GROUP(A_, B_)                                                                    \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n) \
      acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[m], B_[n], acc[m][n], 0, 0, 0);
      if constexpr (NPL == 3) {
        bf16x8 am[MT], bm[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) am[m] = *reinterpret_cast<const bf16x8*>(sa_hi + BM * 64 + fr_a + m * 2048 + choff);
#pragma unroll
        for (int n = 0; n < NT; ++n) bm[n] = *reinterpret_cast<const bf16x8*>(sb_hi + BN * 64 + fr_b + n * 2048 + choff);
        SRN_WS_GROUP(al, bh)
        SRN_WS_GROUP(ah, bl)
        SRN_WS_GROUP(am, bm)
        SRN_WS_GROUP(am, bh)
        SRN_WS_GROUP(ah, bm)
        SRN_WS_GROUP(ah, bh)
      } else {
        SRN_WS_GROUP(al, bh)
        SRN_WS_GROUP(ah, bl)
        SRN_WS_GROUP(ah, bh)
      }
#undef SRN_WS_GROUP
    }
  };
  __syncthreads();  // stage 0 visible
  int step = 0;
  for (; step + 2 <= n_steps; step += 2) {
    compute(0);
    __syncthreads();
    compute(1);
    __syncthreads();
  }
  if (step < n_steps) {
    compute(0);
    __syncthreads();
  }
  conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

template <int NPL, int ACT, bool WPL>
int launch_ws3(const SrnConvParams& p, hipStream_t stream) {
  constexpr int SMEM = 2 * (BM + BN) * 64 * NPL;
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_ws_kernel<NPL, ACT, WPL>), SMEM)) return e;
  const int m_tiles = (p.T_out + BM - 1) / BM;
  const int n_tiles = (p.N + BN - 1) / BN;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m_tiles * n_tiles;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_ws: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL((conv_ws_kernel<NPL, ACT, WPL>), dim3((unsigned)blocks), dim3(512), SMEM, stream, p, m_tiles,
                     n_tiles);
  SRN_CHECK_LAUNCH();
  return 1;
}

template <int NPL, int ACT>
int launch_ws2(const SrnConvParams& p, bool wpl, hipStream_t stream) {
  return wpl ? launch_ws3<NPL, ACT, true>(p, stream) : launch_ws3<NPL, ACT, false>(p, stream);
}

template <int NPL>
int launch_ws(const SrnConvParams& p, bool wpl, hipStream_t stream) {
  if (p.pro_act == SRN_ACT_NONE) return launch_ws2<NPL, SRN_ACT_NONE>(p, wpl, stream);
  if (p.pro_act == SRN_ACT_LEAKY) return launch_ws2<NPL, SRN_ACT_LEAKY>(p, wpl, stream);
  return launch_ws2<NPL, -1>(p, wpl, stream);
}

}  // namespace

// Returns 1 if the launch was handled, 0 if the shape is not eligible, < 0 on error (same contract as
// srn_conv_fast_try).  Takes the 128 x 128 tile of the two split modes only.
int srn_conv_ws_try(const SrnConvParams& p, int tile, hipStream_t stream) {
  if (tile != 1 || p.w_nmajor) return 0;
  if (p.C_in % BK != 0 || p.C_in0 % BK != 0) return 0;
  const bool x6 = p.precision == SRN_PREC_BF16X6;
  if (p.precision != SRN_PREC_BF16X3 && !x6) return 0;
  const bool wpl = p.w_hi != nullptr && (!x6 || p.w_lo != nullptr) && p.w_bs == 0 && p.w_hs == 0;
  if (!wpl && (p.C_w != p.C_in || p.ldw < p.n_taps * p.C_in)) return 0;
  return x6 ? launch_ws<3>(p, wpl, stream) : launch_ws<2>(p, wpl, stream);
}
