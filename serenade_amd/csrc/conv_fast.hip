// conv_fast.hip -- lean split-bf16 implicit-GEMM kernel for the regular shapes of the path
// (C_in a multiple of 32, k-major weights): the same contraction, LDS image, MFMA sequence and epilogue as
// conv_gemm.hip's split-bf16 path, with the per-step instruction count cut from ~420 to ~150 per wave:
//   * measured (tools/ksweep.py on timing-only builds): the generic kernel is ISSUE-bound, not memory- or
//     MFMA-bound -- all-L2-hit operands do not speed it up, removing the global loads' address arithmetic does;
//   * operand cursors: per-lane row pointers are set up once per (tap, input tensor) and bumped by a constant
//     per 32-channel step; padded / masked rows point at a zero page with bump 0 (no selects after the load);
//   * static weights arrive pre-split (bf16 hi|lo planes, one 128-B line per row per 32-channel step, packed at
//     load time by ops.weight_planes): 16-B loads go to LDS untouched, only activations are split in the loop;
//   * the split itself is 2.5 VALU per element (v_cvt_pk_bf16_f32 on pairs, v_pk_add_f32).
// The same loop runs the exact-fp32 mode (FCfg<..., PREC = 0>: fp32 LDS image, v_mfma_f32_32x32x2_f32; 131 TFLOP/s
// sustained on 10240 x 2048 x 2048 = 0.83 of the fp32 MFMA peak).  Shapes it does not take (ragged channels, n-major B)
// stay on conv_gemm.hip.
//
// Measured and NOT adopted (A/B or timing-only builds, 10240 x 2048 x 2048 split-bf16, 351 TFLOP/s sustained baseline):
// v_mfma_f32_16x16x32_bf16 issue +4 %; A operand pre-split as well +2.7 %; half the LDS fragment reads +0 %; one LDS
// stage at three workgroups per CU -1 %; 128-B pad between the B planes (no ds_write_b128 bank conflicts) -2.7 %;
// persistent workgroups prefetching the next tile before the epilogue -17 % (spills); 256 x 256 tile on 8 waves
// +6.5 % at K = 2048 but +1 % at K = 512, where every wide-N GEMM of the path lives.  The card runs the loop at its
// 1370 W cap (1.93 GHz).
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int BK = 32;

// 256 B of zeros: the source of every padded / masked operand row
__device__ __attribute__((aligned(256))) float g_zero_page[64];


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// x = hi + lo, both round-to-nearest bf16; 5 VALU per pair
__device__ __forceinline__ void split_pair(const float a, const float b, unsigned& hi, unsigned& lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2 l = v - hf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
}

// x = hi + mid + lo EXACTLY (three round-to-nearest bf16 of successive exact residuals: 8 + 8 + 8 significand bits
// cover fp32's 24; x - hi and (x - hi) - mid are exact in fp32)
__device__ __forceinline__ void split3_pair(const float a, const float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2 r1 = v - hf;
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  f32x2 mf;
  mf.x = __builtin_bit_cast(float, mid << 16);
  mf.y = __builtin_bit_cast(float, mid & 0xffff0000u);
  const f32x2 r2 = r1 - mf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// NSTAGE: LDS stages.  2: tile s+1 is split / written while tile s is multiplied.  1: the two phases alternate
// between barriers inside a workgroup and a third co-resident workgroup (32 KB each) supplies the overlap.
// PREC: 1 = split-bf16 (3 x v_mfma_f32_32x32x16_bf16 per product), 0 = exact fp32 (v_mfma_f32_32x32x2_f32); the fp32
// LDS image is [rows][32 floats + 4 pad] (144-B rows: conflict-free ds_read_b128 of 16 rows), A then B.
// PREC 2 = bf16x6, fp32-FAITHFUL emulation: operands split exactly into (hi, mid, lo) bf16 planes, a product is the six
// MFMAs lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi (every bf16 x bf16 product is exact in the fp32 accumulate);
// the dropped mid*lo, lo*mid, lo*lo terms are <= 2^-26 of the product -- a quarter of fp32's own rounding unit -- at
// 6 / 16 of the fp32-MFMA cost.  LDS: three 64-B-row planes per operand.
template <int BM_, int BN_, int WM_, int WN_, int NSTAGE_ = 2, int PREC_ = 1>
struct FCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_, PREC = PREC_;
  static constexpr int MT = WM / 32, NT = WN / 32;
  static constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  static constexpr int A_LD = BM / 32, B_LD = BN / 32;  // 16-B loads per thread per step
  static constexpr int NPL = PREC == 2 ? 3 : 2;         // bf16 planes per operand
  static constexpr int B_LD2 = PREC == 2 ? BN / 64 : 0;  // bf16x6: extra 16-B loads per thread for the lo weight plane
  static_assert(PREC != 2 || BN % 64 == 0 || BN == 32, "lo-plane pieces must divide over the workgroup");
  static constexpr int STAGE = (BM + BN) * (PREC ? 64 * NPL : 144);  // split: [A planes | B planes], 64 B per row
  static constexpr int SMEM_BYTES = NSTAGE * STAGE;
  // workgroups per CU the kernel is compiled for (register budget 512 / waves per SIMD): three where the LDS allows
  // (two at most in bf16x6: 3 x (MT + NT) fragments + the accumulators do not fit 168 registers)
  static constexpr int MIN_BLOCKS = SMEM_BYTES * 2 > 160 * 1024 ? 1 : (SMEM_BYTES * 3 <= 160 * 1024 && PREC != 2 ? 3 : 2);
};

// ACT: SRN_ACT_NONE / SRN_ACT_LEAKY compile-time, -1 = run-time p.pro_act (SiLU / Mish).
// WPL: B operand is the pre-split weight plane image (p.w_hi); otherwise fp32 rows split in the loop (Q K^T).

template <class C, int ACT, bool WPL>
__global__ __launch_bounds__(256, C::MIN_BLOCKS) void conv_fast_kernel(const SrnConvParams p, const int m_tiles,
                                                           const int n_tiles, const int ksplit) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int logical = xcd_logical_block();
  // split-K (ksplit > 1): the grid holds ksplit copies of the tile set, slice-major; a block contracts only its slice
  // of the (tap, channel) steps and stores raw partial sums to the workspace (conv_splitk.hip reduces them).
  const int tiles_all = p.n_batch * p.n_head * m_tiles * n_tiles;
  const int slice = logical / tiles_all;
  logical -= slice * tiles_all;
  int z, mt_i, nt_i;
  tile_coords(logical, m_tiles, n_tiles, z, mt_i, nt_i);
  const int zb = z / p.n_head;
  const int zh = z - zb * p.n_head;
  const int t0 = mt_i * BM;
  const int n0 = nt_i * BN;

  const float* in0 = p.in0 + (int64_t)zb * p.in0_bs + (int64_t)zh * p.in0_hs;
  const float* in1 = p.in1 ? p.in1 + (int64_t)zb * p.in1_bs : nullptr;
  const int T_in = p.T_in;
  int len_in = T_in;
  if (p.len_in) len_in = min(p.len_in[zb], T_in);

  const int cpt = p.C_in / BK;   // 32-channel steps per tap
  const int cp0 = p.C_in0 / BK;  // ... of which from in0 (== cpt without a concat input)
  const int steps_all = p.n_taps * cpt;  // 32-deep steps of the whole contraction
  const int per_slice = (steps_all + ksplit - 1) / ksplit;
  const int s_begin = slice * per_slice;
  const int n_steps = max(0, min(steps_all, s_begin + per_slice) - s_begin);  // steps of this block

  // ---- operand cursors (run ahead of the MFMA phase; only load() touches them)
  const int c4 = tid & 7;     // 16-B piece of the 128-B line
  const int lrow = tid >> 3;  // tile rows lrow + 32 i
  int a_tb[C::A_LD];          // input row of the tile row before the tap offset
#pragma unroll
  for (int i = 0; i < C::A_LD; ++i) a_tb[i] = min(t0 + lrow + 32 * i, p.T_out - 1) * p.in_stride;
  const float* aptr[C::A_LD];
  int abump[C::A_LD];  // floats per step: 32, or 0 on the zero page
  int cur_tap = 0, cur_seg = 0, left = 0;
  auto a_setup = [&](const int tap, const int seg) {
    const float* src = seg == 0 ? in0 : in1;
    const int ld = seg == 0 ? p.ld_in0 : p.ld_in1;
    const int toff = p.tap_off[tap];
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      int ti = a_tb[i] + toff;
      if (p.pad_reflect) {  // 2: mirror at the item's own end (ragged batches), else at the tensor's end
        const int T_ref = p.pad_reflect == 2 ? len_in : T_in;
        if (ti < 0) ti = -ti;
        if (ti >= T_ref) ti = 2 * (T_ref - 1) - ti;
      }
      const bool ok = ti >= 0 && ti < len_in;
      aptr[i] = ok ? src + (int64_t)ti * ld + c4 * 4 : g_zero_page + c4 * 4;
      abump[i] = ok ? BK : 0;
    }
    left = seg == 0 ? cp0 : cpt - cp0;
  };
  {
    // first step of this block: (tap, input tensor, 32-channel chunk inside it)
    cur_tap = s_begin / cpt;
    const int within = s_begin - cur_tap * cpt;
    cur_seg = within >= cp0 ? 1 : 0;
    const int chunk = cur_seg ? within - cp0 : within;
    a_setup(min(cur_tap, p.n_taps - 1), cur_seg);
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) aptr[i] += abump[i] * chunk;
    left -= chunk;
  }

  const float* bptr[C::B_LD];  // WPL: 16-B piece c4 of the row's current 128-B plane line; else fp32 row + 4 c4
#pragma unroll
  for (int i = 0; i < C::B_LD; ++i) {
    const int n = min(n0 + lrow + 32 * i, p.N - 1);  // columns >= N are computed on a clamped row, never stored
    if constexpr (WPL)
      bptr[i] = reinterpret_cast<const float*>(p.w_hi) + ((int64_t)n * steps_all * 32 + c4 * 4);
    else
      bptr[i] = p.w + (int64_t)zb * p.w_bs + (int64_t)zh * p.w_hs + (int64_t)n * p.ldw + c4 * 4;
  }

#pragma unroll
  for (int i = 0; i < C::B_LD; ++i) bptr[i] += (int64_t)s_begin * BK;
  int bbump = BK;  // floats per step (0 once the cursor is parked)
  // bf16x6 with pre-split weights: the (hi | mid) planes arrive through bptr like bf16x3's (hi | lo); the lo plane is a
  // second image [N][steps][32 bf16] (p.w_lo) whose 64-B lines are 4 pieces: thread -> (row tid / 4 + 64 j, piece tid % 4)
  constexpr int B_LD2 = (WPL && C::PREC == 2) ? (C::B_LD2 > 0 ? C::B_LD2 : 1) : 0;
  constexpr bool LO_ALL = C::BN >= 64;  // BN = 32: only threads 0..127 carry a piece
  const float* cptr[B_LD2 > 0 ? B_LD2 : 1];
  const int c_q4 = tid & 3, c_row = tid >> 2;
  if constexpr (B_LD2 > 0) {
#pragma unroll
    for (int j = 0; j < B_LD2; ++j) {
      const int n = min(n0 + min(c_row + 64 * j, C::BN - 1), p.N - 1);
      cptr[j] = reinterpret_cast<const float*>(p.w_lo) + ((int64_t)n * steps_all + s_begin) * 16 + c_q4 * 4;
    }
  }
  struct Regs {
    float4 pa[C::A_LD];
    float4 pb[C::B_LD];
    float4 pc[B_LD2 > 0 ? B_LD2 : 1];
  };
  // issue-only: nothing here consumes a loaded value
  auto load_issue = [&](Regs& R) {
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      R.pa[i] = *reinterpret_cast<const float4*>(aptr[i]);
      aptr[i] += abump[i];
    }
#pragma unroll
    for (int i = 0; i < C::B_LD; ++i) {
      R.pb[i] = *reinterpret_cast<const float4*>(bptr[i]);
      bptr[i] += bbump;
    }
    if constexpr (B_LD2 > 0) {
#pragma unroll
      for (int j = 0; j < B_LD2; ++j) {
        R.pc[j] = *reinterpret_cast<const float4*>(cptr[j]);
        cptr[j] += bbump >> 1;  // 64-B lines
      }
    }
  };
  auto cursor_advance = [&]() {
    if (--left == 0) {  // wave-uniform, once per (tap, input tensor)
      if (cur_seg == 0 && cp0 < cpt) {
        cur_seg = 1;
      } else {
        cur_seg = 0;
        ++cur_tap;
      }
      if (cur_tap < p.n_taps) {
        a_setup(cur_tap, cur_seg);
      } else {
        // past the last step: the pipeline issues up to two more (unconditional) loads; park the cursors on
        // valid memory.  Unconditional loads keep hipcc's s_waitcnt vmcnt(N) counted -- with a load under an
        // `if` it must assume the newer loads may not exist and waits vmcnt(0), i.e. for the loads just issued.
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i) {
          aptr[i] = g_zero_page + c4 * 4;
          abump[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) bptr[i] -= BK;
        if constexpr (B_LD2 > 0) {
#pragma unroll
          for (int j = 0; j < B_LD2; ++j) cptr[j] -= BK / 2;
        }
        bbump = 0;
        left = 1 << 30;
      }
    }
  };

  const int pro_act = p.pro_act;
  const float pro_slope = p.pro_slope;
  const int st_off = bf_off(lrow, c4 * 4);          // A (and fp32 B) rows: 8-B slot of this thread's float4
  const int stb_off = bf_off(lrow, (c4 & 3) * 8);   // plane B rows: 16-B slot of this thread's piece
  constexpr int NPL = C::NPL;
  const int stc_off = bf_off(c_row, c_q4 * 8);  // bf16x6 lo weight plane: 16-B slot of this thread's piece
  auto store = [&](const int stage, Regs& R) {
    // planes: A [hi | (mid) | lo] then B [hi | (mid) | lo]; "lo" below is the LAST plane, "mid" exists for NPL = 3
    unsigned char* sa_hi = smem_f + stage * C::STAGE;
    unsigned char* sa_lo = sa_hi + (NPL - 1) * BM * 64;
    unsigned char* sb_hi = sa_hi + NPL * BM * 64;
    unsigned char* sb_lo = sb_hi + (NPL - 1) * BN * 64;
    if constexpr (C::PREC == 0) {
      float* a32 = reinterpret_cast<float*>(smem_f + stage * C::STAGE) + lrow * 36 + c4 * 4;
      float* b32 = a32 + BM * 36;
#pragma unroll
      for (int i = 0; i < C::A_LD; ++i) {
        float4 v = R.pa[i];
        if constexpr (ACT == SRN_ACT_LEAKY) {
          v.x = v.x > 0.f ? v.x : v.x * pro_slope;
          v.y = v.y > 0.f ? v.y : v.y * pro_slope;
          v.z = v.z > 0.f ? v.z : v.z * pro_slope;
          v.w = v.w > 0.f ? v.w : v.w * pro_slope;
        } else if constexpr (ACT < 0) {
          v.x = srn_act(v.x, pro_act, pro_slope);
          v.y = srn_act(v.y, pro_act, pro_slope);
          v.z = srn_act(v.z, pro_act, pro_slope);
          v.w = srn_act(v.w, pro_act, pro_slope);
        } else {
          asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));  // keep the staged set out of scratch
        }
        *reinterpret_cast<float4*>(a32 + i * 32 * 36) = v;
      }
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) {
        float4 v = R.pb[i];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        *reinterpret_cast<float4*>(b32 + i * 32 * 36) = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      float4 v = R.pa[i];
      if constexpr (ACT == SRN_ACT_LEAKY) {
        v.x = v.x > 0.f ? v.x : v.x * pro_slope;
        v.y = v.y > 0.f ? v.y : v.y * pro_slope;
        v.z = v.z > 0.f ? v.z : v.z * pro_slope;
        v.w = v.w > 0.f ? v.w : v.w * pro_slope;
      } else if constexpr (ACT < 0) {
        v.x = srn_act(v.x, pro_act, pro_slope);
        v.y = srn_act(v.y, pro_act, pro_slope);
        v.z = srn_act(v.z, pro_act, pro_slope);
        v.w = srn_act(v.w, pro_act, pro_slope);
      }
      if constexpr (NPL == 3) {
        uint2 hi, mid, lo;
        split3_pair(v.x, v.y, hi.x, mid.x, lo.x);
        split3_pair(v.z, v.w, hi.y, mid.y, lo.y);
        *reinterpret_cast<uint2*>(sa_hi + st_off + i * 2048) = hi;
        *reinterpret_cast<uint2*>(sa_hi + BM * 64 + st_off + i * 2048) = mid;
        *reinterpret_cast<uint2*>(sa_lo + st_off + i * 2048) = lo;
      } else {
        uint2 hi, lo;
        split_pair(v.x, v.y, hi.x, lo.x);
        split_pair(v.z, v.w, hi.y, lo.y);
        *reinterpret_cast<uint2*>(sa_hi + st_off + i * 2048) = hi;
        *reinterpret_cast<uint2*>(sa_lo + st_off + i * 2048) = lo;
      }
    }
    if constexpr (WPL) {
      // (hi | lo) of bf16x3, (hi | mid) of bf16x6: 128-B lines, pieces 0-3 -> first plane, 4-7 -> second
      unsigned char* dst = (c4 < 4 ? sb_hi : sb_hi + BN * 64) + stb_off;
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) *reinterpret_cast<float4*>(dst + i * 2048) = R.pb[i];
      if constexpr (B_LD2 > 0) {
        if (LO_ALL || c_row < C::BN) {
#pragma unroll
          for (int j = 0; j < B_LD2; ++j) {
            float4 v = R.pc[j];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            *reinterpret_cast<float4*>(sb_lo + stc_off + j * 4096) = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < C::B_LD; ++i) {
        const float4 v = R.pb[i];
        if constexpr (NPL == 3) {
          uint2 hi, mid, lo;
          split3_pair(v.x, v.y, hi.x, mid.x, lo.x);
          split3_pair(v.z, v.w, hi.y, mid.y, lo.y);
          *reinterpret_cast<uint2*>(sb_hi + st_off + i * 2048) = hi;
          *reinterpret_cast<uint2*>(sb_hi + BN * 64 + st_off + i * 2048) = mid;
          *reinterpret_cast<uint2*>(sb_lo + st_off + i * 2048) = lo;
        } else {
          uint2 hi, lo;
          split_pair(v.x, v.y, hi.x, lo.x);
          split_pair(v.z, v.w, hi.y, lo.y);
          *reinterpret_cast<uint2*>(sb_hi + st_off + i * 2048) = hi;
          *reinterpret_cast<uint2*>(sb_lo + st_off + i * 2048) = lo;
        }
      }
    }
  };

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

  // Split-bf16 modes multiply on v_mfma_f32_16x16x32_bf16 (round 2 measured it clocking ~6 % higher than 32x32x16 at the
  // board's power cap): the wave's tile is (2 MT) x (2 NT) blocks of 16 x 16, one MFMA per block, plane pair and 32-deep
  // step.  Lane l feeds row (l & 15), k chunk (l >> 4) of a block.  Block rows are taken in the order rho(j) = 4 pi(j >> 2)
  // + (j & 3), pi = (0, 3, 2, 1): with the image's XOR swizzle (chunk ^ row quad) that makes every ds_read_b128 of a
  // fragment conflict-free (identity order is 2-way).  After the loop the 4-register block accumulators are permuted into
  // the 32 x 32 MFMA's C/D layout (two ds_bpermute + a select per register), so epilogue and split-K store are shared.
  constexpr bool M16 = C::PREC != 0;
  constexpr int RB = M16 ? 2 * MT : 1, CB = M16 ? 2 * NT : 1;
  f32x4m acc4[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;
  const int l15 = lane & 15;
  const int q15 = l15 >> 2;
  const int piq = q15 == 1 ? 3 : (q15 == 3 ? 1 : q15);
  const int rho15 = 4 * piq + (l15 & 3);
  const int f16_off = rho15 * 64 + ((((lane >> 4) ^ piq) & 3) << 4);  // byte offset of this lane's fragment inside a block

  auto compute = [&](const int stage) {
    const unsigned char* sa_hi = smem_f + stage * C::STAGE;
    const unsigned char* sa_lo = sa_hi + (NPL - 1) * BM * 64;
    const unsigned char* sb_hi = sa_hi + NPL * BM * 64;
    const unsigned char* sb_lo = sb_hi + (NPL - 1) * BN * 64;
    if constexpr (C::PREC == 0) {
      const float* a = reinterpret_cast<const float*>(smem_f + stage * C::STAGE) + (wm0 + li) * 36 + 4 * lh;
      const float* b = reinterpret_cast<const float*>(smem_f + stage * C::STAGE) + (BM + wn0 + li) * 36 + 4 * lh;
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        float4 af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a + m * 32 * 36 + kk * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float4*>(b + n * 32 * 36 + kk * 8);
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
      return;
    }
    {
      bf16x8 ah[RB], al[RB], bh[CB], bl[CB];
      const unsigned char* pa = sa_hi + wm0 * 64 + f16_off;
      const unsigned char* pb = sb_hi + wn0 * 64 + f16_off;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 1024);
        al[i] = *reinterpret_cast<const bf16x8*>(pa + (NPL - 1) * BM * 64 + i * 1024);
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(pb + j * 1024);
        bl[j] = *reinterpret_cast<const bf16x8*>(pb + (NPL - 1) * BN * 64 + j * 1024);
      }
#define SRN_MFMA_GROUP(A_, B_)                                                                 \
  _Pragma("unroll") for (int i = 0; i < RB; ++i) _Pragma("unroll") for (int j = 0; j < CB; ++j) \
      acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[i], B_[j], acc4[i][j], 0, 0, 0);
      if constexpr (NPL == 3) {
        bf16x8 am[RB], bm[CB];
#pragma unroll
        for (int i = 0; i < RB; ++i) am[i] = *reinterpret_cast<const bf16x8*>(pa + BM * 64 + i * 1024);
#pragma unroll
        for (int j = 0; j < CB; ++j) bm[j] = *reinterpret_cast<const bf16x8*>(pb + BN * 64 + j * 1024);
        // smallest terms first: 2^-16 (lo*hi, hi*lo, mid*mid), 2^-8 (mid*hi, hi*mid), 1 (hi*hi)
        SRN_MFMA_GROUP(al, bh)
        SRN_MFMA_GROUP(ah, bl)
        SRN_MFMA_GROUP(am, bm)
        SRN_MFMA_GROUP(am, bh)
        SRN_MFMA_GROUP(ah, bm)
        SRN_MFMA_GROUP(ah, bh)
      } else {
        SRN_MFMA_GROUP(al, bh)
        SRN_MFMA_GROUP(ah, bl)
        SRN_MFMA_GROUP(ah, bh)
      }
#undef SRN_MFMA_GROUP
    }
  };

  // MFMA phase of tile s and staging of tile s+1 in ONE basic block, interleaved by the scheduler: each MFMA holds
  // the vector issue port for 8 of its 32 cycles, so ~5 VALU / LDS instructions ride along per MFMA.
  // The global loads of tile s+2 ride in the same block (one per ~3 MFMAs): issued back to back ahead of the MFMAs
  // they cost the wave ~565 cycles per step waiting on the address / data path with its MFMA queue empty.
  auto fused = [&](const int cs, const int ss, Regs& Rs, Regs& Rl) {
    compute(cs);
    store(ss, Rs);
    load_issue(Rl);
    constexpr bool F32 = C::PREC == 0;
    constexpr int N_MFMA = MT * NT * (F32 ? 16 : 4 * 3 * (NPL - 1));  // split modes: 4 blocks of 16 x 16 per 32 x 32
    constexpr int N_DSR = F32 ? (MT + NT) * 4 : (MT + NT) * 2 * NPL;  // one b128 per 16-row block and plane
    constexpr int N_LD = C::A_LD + C::B_LD + B_LD2;
    constexpr int SPL = NPL == 3 ? 16 : 10;  // VALU of one float4 split
    constexpr int N_VALU = F32 ? C::A_LD * (ACT == SRN_ACT_NONE ? 0 : 8) + N_LD
                               : C::A_LD * (SPL + (ACT == SRN_ACT_NONE ? 0 : 8)) + (WPL ? 0 : C::B_LD * SPL) + N_LD;
    constexpr int N_DSW = F32 ? C::A_LD + C::B_LD : C::A_LD * NPL + (WPL ? C::B_LD + B_LD2 : C::B_LD * NPL);
    constexpr int VPM = (N_VALU + N_MFMA - 1) / N_MFMA;
    constexpr int DSR0 = F32 ? (MT + NT) : (MT + NT) * NPL;
    __builtin_amdgcn_sched_group_barrier(0x100, DSR0, 0);  // fragments of the first k slice
#pragma unroll
    for (int i = 0; i < N_MFMA; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < N_DSR - DSR0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (i * N_LD / N_MFMA != (i + 1) * N_LD / N_MFMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (i * N_DSW / N_MFMA != (i + 1) * N_DSW / N_MFMA) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  auto load = [&](Regs& R) {
    load_issue(R);
    cursor_advance();
  };


  // 16 x 16 block accumulators -> the 32 x 32 MFMA's C/D layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
  // element (row, col) of a 32 x 32 sub-tile sits in block (row >> 4, col >> 4), register row & 3, lane rho(col & 15) +
  // 16 pi((row & 15) >> 2)
  auto to_32x32 = [&]() {
    if constexpr (M16) {
      const int src_lo = rho15;  // rho(lane & 15)
      const int qa = lh, qb = 2 + lh;  // (row & 15) >> 2 for r >> 2 even / odd
      const int pqa = qa == 1 ? 3 : (qa == 3 ? 1 : qa), pqb = qb == 1 ? 3 : (qb == 3 ? 1 : qb);
      const int addr_a = (src_lo + 16 * pqa) * 4, addr_b = (src_lo + 16 * pqb) * 4;
      const bool right = (lane >> 4) & 1;  // this lane's column is in the right-hand 16 x 16 block
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int bi = 2 * m + (r >> 3);
            const int addr = ((r >> 2) & 1) ? addr_b : addr_a;
            // (the element goes through a scalar first: __builtin_bit_cast applied to the vector element itself
            // picked the wrong register on this hipcc -- tools/experiments/mfma16_check2.hip)
            const float x0 = acc4[bi][2 * n][r & 3], x1 = acc4[bi][2 * n + 1][r & 3];
            const int v0 = __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x0));
            const int v1 = __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x1));
            acc[m][n][r] = __builtin_bit_cast(float, right ? v1 : v0);
          }
    }
  };

  if constexpr (C::NSTAGE == 1) {
    // One LDS stage, one register set: [split / write tile s | barrier | loads of tile s+1 issued, MFMAs of tile s |
    // barrier].  Nothing overlaps inside the workgroup; three co-resident workgroups overlap each other.
    Regs R;
    load(R);
    for (int s = 0; s < n_steps; ++s) {
      store(0, R);
      __syncthreads();
      load_issue(R);
      compute(0);
      cursor_advance();
      __syncthreads();
    }
    to_32x32();
    if (ksplit > 1) splitk_store<MT, NT>(p, acc, slice, z, t0, n0, wm0, wn0, lane);
    else conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
    return;
  }

  // Pipeline: LDS double-buffered, two register sets; the loads of tile s+2 are issued inside the MFMA phase of
  // tile s, whose basic block also splits / writes tile s+1.
  Regs R0, R1;
  load(R0);
  store(0, R0);
  load(R0);
  __syncthreads();
  int step = 0;
  // invariant: LDS[0] holds tile `step`; R0 holds the raw tile step+1 (or a parked dummy)
  for (; step + 2 < n_steps; step += 2) {
    fused(0, 1, R0, R1);
    cursor_advance();
    __syncthreads();
    fused(1, 0, R1, R0);
    cursor_advance();
    __syncthreads();
  }
  compute(0);
  if (step + 1 < n_steps) {
    store(1, R0);
    __syncthreads();
    compute(1);
  }

  to_32x32();
  if (ksplit > 1) splitk_store<MT, NT>(p, acc, slice, z, t0, n0, wm0, wn0, lane);
  else conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

template <class C, int ACT, bool WPL>
int launch_fast3(const SrnConvParams& p, hipStream_t stream, const int ksplit) {
  constexpr int SMEM = C::SMEM_BYTES;
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_fast_kernel<C, ACT, WPL>), SMEM)) return e;
  const int m_tiles = (p.T_out + C::BM - 1) / C::BM;
  const int n_tiles = (p.N + C::BN - 1) / C::BN;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m_tiles * n_tiles * ksplit;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_fast: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL((conv_fast_kernel<C, ACT, WPL>), dim3((unsigned)blocks), dim3(256), SMEM, stream, p, m_tiles,
                     n_tiles, ksplit);
  SRN_CHECK_LAUNCH();
  return 1;
}

template <class C, int ACT>
int launch_fast2(const SrnConvParams& p, bool wpl, hipStream_t stream, int ks) {
  if constexpr (C::PREC == 0) return launch_fast3<C, ACT, false>(p, stream, ks);  // fp32: plain weight rows
  else return wpl ? launch_fast3<C, ACT, true>(p, stream, ks) : launch_fast3<C, ACT, false>(p, stream, ks);
}

template <class C>
int launch_fast(const SrnConvParams& p, bool wpl, hipStream_t stream, int ks = 1) {
  if (p.pro_act == SRN_ACT_NONE) return launch_fast2<C, SRN_ACT_NONE>(p, wpl, stream, ks);
  if (p.pro_act == SRN_ACT_LEAKY) return launch_fast2<C, SRN_ACT_LEAKY>(p, wpl, stream, ks);
  return launch_fast2<C, -1>(p, wpl, stream, ks);
}

}  // namespace

// Returns 1 if the launch was handled, 0 if the shape is not eligible (caller falls back to the generic kernel),
// < 0 on error.  `p` has been validated and defaulted by srn_conv_gemm.
int srn_conv_fast_try(const SrnConvParams& p, int tile, hipStream_t stream, int ksplit) {
  if (p.w_nmajor) return 0;
  if (p.C_in % BK != 0 || p.C_in0 % BK != 0) return 0;
  const bool x6 = p.precision == SRN_PREC_BF16X6;
  const bool f32 = p.precision != SRN_PREC_BF16X3 && !x6;
  const bool wpl = !f32 && p.w_hi != nullptr && (!x6 || p.w_lo != nullptr) && p.w_bs == 0 && p.w_hs == 0;
  if (!wpl) {
    // fp32 B rows walked contiguously over (tap, channel): needs the packed [tap][C_in] row layout, all of it live
    if (p.C_w != p.C_in || p.ldw < p.n_taps * p.C_in) return 0;
  }
  if (ksplit > 1) {  // split-K launches always take the 64 x 64 tile (they exist because the grid is small)
    if (f32) return launch_fast<FCfg<64, 64, 32, 32, 2, 0>>(p, false, stream, ksplit);
    if (x6) return launch_fast<FCfg<64, 64, 32, 32, 2, 2>>(p, wpl, stream, ksplit);
    return launch_fast<FCfg<64, 64, 32, 32>>(p, wpl, stream, ksplit);
  }
  if (f32) {
    switch (tile) {
      case 1: return launch_fast<FCfg<128, 128, 64, 64, 2, 0>>(p, false, stream);
      case 2: return launch_fast<FCfg<128, 64, 32, 64, 2, 0>>(p, false, stream);
      case 3: return launch_fast<FCfg<64, 128, 32, 64, 2, 0>>(p, false, stream);
      case 4: return launch_fast<FCfg<64, 64, 32, 32, 2, 0>>(p, false, stream);
      case 5: return launch_fast<FCfg<128, 32, 32, 32, 2, 0>>(p, false, stream);
      case 6: return launch_fast<FCfg<128, 128, 64, 64, 1, 0>>(p, false, stream);  // single LDS stage
      case 7: return launch_fast<FCfg<64, 64, 32, 32, 1, 0>>(p, false, stream);
      case 8: return launch_fast<FCfg<128, 64, 32, 64, 1, 0>>(p, false, stream);
      case 9: return launch_fast<FCfg<64, 128, 32, 64, 1, 0>>(p, false, stream);
      default: return 0;
    }
  }
  if (x6) {
    switch (tile) {
      // one LDS stage (48 KB, three workgroups per CU); the two-stage form (96 KB, one per CU) measured 0-40 % slower
      case 1: return launch_fast<FCfg<128, 128, 64, 64, 1, 2>>(p, wpl, stream);
      case 2: return launch_fast<FCfg<128, 64, 32, 64, 2, 2>>(p, wpl, stream);
      case 3: return launch_fast<FCfg<64, 128, 32, 64, 2, 2>>(p, wpl, stream);
      case 4: return launch_fast<FCfg<64, 64, 32, 32, 2, 2>>(p, wpl, stream);
      case 5: return launch_fast<FCfg<128, 32, 32, 32, 2, 2>>(p, wpl, stream);
      case 7: return launch_fast<FCfg<64, 64, 32, 32, 1, 2>>(p, wpl, stream);
      case 8: return launch_fast<FCfg<128, 64, 32, 64, 1, 2>>(p, wpl, stream);
      case 9: return launch_fast<FCfg<64, 128, 32, 64, 1, 2>>(p, wpl, stream);
      default: return 0;
    }
  }
  switch (tile) {
    case 1: return launch_fast<FCfg<128, 128, 64, 64>>(p, wpl, stream);
    case 2: return launch_fast<FCfg<128, 64, 32, 64>>(p, wpl, stream);
    case 3: return launch_fast<FCfg<64, 128, 32, 64>>(p, wpl, stream);
    case 4: return launch_fast<FCfg<64, 64, 32, 32>>(p, wpl, stream);
    case 5: return launch_fast<FCfg<128, 32, 32, 32>>(p, wpl, stream);
    case 6: return launch_fast<FCfg<128, 128, 64, 64, 1>>(p, wpl, stream);  // single LDS stage, 3 workgroups / CU
    case 7: return launch_fast<FCfg<64, 64, 32, 32, 1>>(p, wpl, stream);
    case 8: return launch_fast<FCfg<128, 64, 32, 64, 1>>(p, wpl, stream);
    case 9: return launch_fast<FCfg<64, 128, 32, 64, 1>>(p, wpl, stream);
    default: return 0;
  }
}

