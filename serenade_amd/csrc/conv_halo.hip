// conv_halo.hip — receptive-field ("halo") variant of the implicit-GEMM conv for stride-1, multi-tap Conv1d in
// split-bf16 precision (the Block1D k3 convs of the UNet, every k3/k7/k11 dilated conv of HiFi-GAN, the 2-tap
// transposed-conv phases).
//
// conv_gemm.hip re-gathers (and re-splits into hi/lo bf16) the shifted A tile for every tap.  Here the A tile of one
// 32-channel chunk is staged ONCE with its halo -- rows [t0 + min_off, t0 + BM + max_off) -- and every tap reads the
// same LDS image at a row shift; only the small weight tile changes per step.  A-side L2 traffic, fp32->bf16 split
// VALU work and LDS writes drop by the tap count (3x for the UNet, up to 11x for the vocoder), which is what the
// short split-bf16 MFMA phase (24 x 32 cycles per 128x128x32 step) needs.
//
// LDS (bytes): A halo image [2 stages][hi | lo][BM + HALO_MAX rows][64 B, XOR-swizzled], weights [2][hi | lo][BN][64 B].
#include "conv_common.h"

namespace {

constexpr int BK = 32;
constexpr int HALO_MAX = 52;  // (k-1)*dilation of the widest conv on the path: k 11, d 5 -> 50

template <int BM_, int BN_, int WM_, int WN_>
struct HCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int MT = WM / 32, NT = WN / 32;
  static constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  static constexpr int HR = BM + HALO_MAX;            // rows of the staged A image
  static constexpr int A_LD = (HR * 8 + 255) / 256;   // float4 loads per thread per chunk
  static constexpr int B_LD = BN / 32;                // float4 loads per thread per tap
  // taps per pipeline step: small wave tiles have only 6-12 MFMAs per tap, so several taps' weight tiles are
  // staged together and multiplied between two barriers (>= 24 MFMAs per wave per barrier)
  static constexpr int TG = 1;  // measured: grouping 2-4 taps per barrier does not help (and hurts the 64x128 tile)
  static constexpr int A_STAGE = HR * 128;            // hi + lo planes
  static constexpr int B_TAP = BN * 128;
  static constexpr int B_STAGE = TG * B_TAP;
  static constexpr int SMEM_BYTES = 2 * (A_STAGE + B_STAGE);
};

template <class C, int ACT>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const SrnConvParams p, const int m_tiles,
                                                           const int n_tiles, const int min_off, const int halo) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  unsigned char* sA = smem_h;
  unsigned char* sB = smem_h + 2 * C::A_STAGE;

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

  constexpr int TG = C::TG;
  const int n_taps = p.n_taps;
  const int n_chunks = (p.C_in + BK - 1) / BK;
  const int ngpc = (n_taps + TG - 1) / TG;  // tap groups per chunk
  const int n_steps = ngpc * n_chunks;
  const int hr = BM + halo;  // rows actually needed

  const int c4 = tid & 7;
  const int lrow = tid >> 3;
  const float pro_slope = p.pro_slope;

  float4 pa[C::A_LD];
  float4 pb[TG][C::B_LD];
  unsigned a_ok = 0, b_ok = 0;

  // ---- A: one 32-channel chunk of the halo rows (issue only)
  auto load_a = [&](const int chunk) {
    const int ch = chunk * BK + c4 * 4;
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
    a_ok = 0;
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      const int row = lrow + 32 * i;
      if (row < hr) {  // (whole waves skip the unused tail of the image)
        int ti = t0 + min_off + row;
        const bool ok = cok && ti >= 0 && ti < len_in;
        a_ok |= (ok ? 1u : 0u) << i;
        ti = min(max(ti, 0), T_in - 1);
        pa[i] = *reinterpret_cast<const float4*>(src + (int64_t)ti * ld + c);
      }
    }
  };
  auto stage_a = [&](const int stage) {
    unsigned char* hi_p = sA + stage * C::A_STAGE;
    unsigned char* lo_p = hi_p + C::HR * 64;
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
      const int row = lrow + 32 * i;
      if (row < hr) {
        float4 v = sel4((a_ok >> i) & 1u, pa[i]);
        if constexpr (ACT == SRN_ACT_LEAKY) v = leaky4(v, pro_slope);
        bf16x4 hi, lo;
        split4(v, hi, lo);
        const int off = bf_off(row, c4 * 4);
        *reinterpret_cast<bf16x4*>(hi_p + off) = hi;
        *reinterpret_cast<bf16x4*>(lo_p + off) = lo;
      }
    }
  };
  // ---- B: the weight tiles of one step = TG consecutive taps of one chunk
  auto load_b = [&](const int step) {
    const int chunk = step / ngpc;
    const int tap0 = (step - chunk * ngpc) * TG;
    const int ch = chunk * BK + c4 * 4;
    const bool kok = ch < p.C_w;
    b_ok = 0;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
      if (tap0 + tt < n_taps) {
        const int64_t kcol = (int64_t)(tap0 + tt) * p.C_in + (kok ? ch : 0);
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
          int n = n0 + lrow + 32 * i;
          const bool ok = kok && n < p.N;
          b_ok |= (ok ? 1u : 0u) << (tt * C::B_LD + i);
          n = min(n, p.N - 1);
          pb[tt][i] = *reinterpret_cast<const float4*>(wgt + (int64_t)n * p.ldw + kcol);
        }
      }
    }
  };
  auto stage_b = [&](const int stage, const int step) {
    const int chunk = step / ngpc;
    const int tap0 = (step - chunk * ngpc) * TG;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
      if (tap0 + tt < n_taps) {
        unsigned char* hi_p = sB + stage * C::B_STAGE + tt * C::B_TAP;
        unsigned char* lo_p = hi_p + BN * 64;
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
          bf16x4 hi, lo;
          split4(sel4((b_ok >> (tt * C::B_LD + i)) & 1u, pb[tt][i]), hi, lo);
          const int off = bf_off(lrow + 32 * i, c4 * 4);
          *reinterpret_cast<bf16x4*>(hi_p + off) = hi;
          *reinterpret_cast<bf16x4*>(lo_p + off) = lo;
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

  auto compute = [&](const int a_stage, const int b_stage, const int tt, const int shift) {
    const unsigned char* a_hi = sA + a_stage * C::A_STAGE;
    const unsigned char* a_lo = a_hi + C::HR * 64;
    const unsigned char* b_hi = sB + b_stage * C::B_STAGE + tt * C::B_TAP;
    const unsigned char* b_lo = b_hi + BN * 64;
    const int swb = (li >> 2) & 3;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = wm0 + m * 32 + li + shift;  // shifted row of the halo image
        const int o = row * 64 + ((((kk * 2 + lh) ^ (row >> 2)) & 3) << 4);
        ah[m] = *reinterpret_cast<const bf16x8*>(a_hi + o);
        al[m] = *reinterpret_cast<const bf16x8*>(a_lo + o);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int o = (wn0 + n * 32 + li) * 64 + ((((kk * 2 + lh) ^ swb) & 3) << 4);
        bh[n] = *reinterpret_cast<const bf16x8*>(b_hi + o);
        bl[n] = *reinterpret_cast<const bf16x8*>(b_lo + o);
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
  };

  // ---- pipeline: A image double-buffered per chunk (prefetched a whole chunk ahead), weight tiles of TG taps
  //      double-buffered per step
  load_a(0);
  load_b(0);
  stage_a(0);
  stage_b(0, 0);
  if (n_chunks > 1) load_a(1);
  __syncthreads();
  int chunk = 0, grp = 0;
  for (int step = 0; step < n_steps; ++step) {
    const bool more = step + 1 < n_steps;
    if (more) load_b(step + 1);
    const int tap0 = grp * TG;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt)
      if (tap0 + tt < n_taps) compute(chunk & 1, step & 1, tt, p.tap_off[tap0 + tt] - min_off);
    if (more) stage_b((step + 1) & 1, step + 1);
    if (++grp == ngpc) {
      grp = 0;
      ++chunk;
      if (chunk < n_chunks) {
        stage_a(chunk & 1);  // loaded a whole chunk ago
        if (chunk + 1 < n_chunks) load_a(chunk + 1);
      }
    }
    __syncthreads();
  }

  conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

template <class C, int ACT>
int launch_halo(const SrnConvParams& p, int min_off, int halo, hipStream_t stream) {
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_halo_kernel<C, ACT>), C::SMEM_BYTES)) return e;
  const int m_tiles = (p.T_out + C::BM - 1) / C::BM;
  const int n_tiles = (p.N + C::BN - 1) / C::BN;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m_tiles * n_tiles;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_halo: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL((conv_halo_kernel<C, ACT>), dim3((unsigned)blocks), dim3(256), C::SMEM_BYTES, stream, p, m_tiles,
                     n_tiles, min_off, halo);
  SRN_CHECK_LAUNCH();
  return 1;
}

template <class C>
int launch_halo_act(const SrnConvParams& p, int min_off, int halo, hipStream_t stream) {
  if (p.pro_act == SRN_ACT_LEAKY) return launch_halo<C, SRN_ACT_LEAKY>(p, min_off, halo, stream);
  return launch_halo<C, SRN_ACT_NONE>(p, min_off, halo, stream);
}

}  // namespace

int srn_conv_halo_try(const SrnConvParams& p, int tile, hipStream_t stream) {
  if (p.precision != SRN_PREC_BF16X3 || p.n_taps < 2 || p.in_stride != 1 || p.pad_reflect || p.w_nmajor || p.geglu)
    return 0;
  if (!(p.pro_act == SRN_ACT_NONE || p.pro_act == SRN_ACT_LEAKY)) return 0;
  int lo = p.tap_off[0], hi = p.tap_off[0];
  for (int i = 1; i < p.n_taps; ++i) {
    lo = p.tap_off[i] < lo ? p.tap_off[i] : lo;
    hi = p.tap_off[i] > hi ? p.tap_off[i] : hi;
  }
  if (hi - lo > HALO_MAX) return 0;
  // Measured on MI355X (tools/opbench.py --bf16x3 [--no-halo]): the halo image pays off when it is reused by many
  // taps of a wide tile (N >= 128: k7 +15..17 %, k11 +22..25 %); for k3 and for the thin N = 32 / 64 tiles, whose
  // steps are latency- not staging-bound, the generic kernel (3 resident blocks per CU) is as fast or faster.
  if (p.no_halo != 2 && !(p.N >= 128 && p.n_taps >= 7)) return 0;
  switch (tile) {
    case 1: return launch_halo_act<HCfg<128, 128, 64, 64>>(p, lo, hi - lo, stream);
    case 2: return launch_halo_act<HCfg<128, 64, 32, 64>>(p, lo, hi - lo, stream);
    case 3: return launch_halo_act<HCfg<64, 128, 32, 64>>(p, lo, hi - lo, stream);
    case 4: return launch_halo_act<HCfg<64, 64, 32, 32>>(p, lo, hi - lo, stream);
    case 5: return launch_halo_act<HCfg<128, 32, 32, 32>>(p, lo, hi - lo, stream);
    default: return 0;
  }
}
