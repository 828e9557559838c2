// resunit.hip -- one HiFi-GAN residual unit (LeakyReLU -> Conv1d k,dil -> LeakyReLU -> Conv1d k,1 -> + x) fused into
// one launch for the thin stages (C = 32 / 64), in both contraction modes.  Reference: HiFiGANResidualBlock.forward,
// serenade/vocoder/layers/residual_block.py:243-258; stage sum / mean of hifigan.py:183-186 in the epilogue.
//
// Why: at 64 / 32 channels a conv moves 2-3 tensors of 252 MB (B = 8, T = 1024) for 24-88 GFLOP -- the unfused unit
// makes five HBM passes (x, xt out, xt in, x again as residual, y) and, on the tiled kernels, re-stages and re-splits
// every input row once per tap.  Here a persistent workgroup walks output tiles; per tile it
//   1. stages the receptive field of the tile (BMI + (k-1) dil rows of x, LeakyReLU applied, split once in bf16x3
//      mode) as an LDS image -- rows were prefetched into registers under the previous tile's MFMAs;
//   2. conv1: every tap reads the image at a row offset; weights arrive through a double-buffered LDS stage of one or
//      two (tap, 32-channel chunk) units, loaded from the L2-resident weight tensor one stage ahead;
//   3. writes bias + LeakyReLU of the BMI intermediate rows back into LDS over the image (zero outside [0, T): conv2
//      zero-pads ITS input), and runs conv2 the same way;
//   4. epilogue: + bias + x (exact fp32 from global) [+ res2] [/ post_div], BMI - (k-1) output rows.
// conv1 is over-computed by (k-1) / BMI (8 % at k = 11, C = 64); HBM traffic is one read of x (+ its L2-hit re-read as
// the residual) and one write of y.  Two workgroups per CU; 4 waves each (wave tile 32 MT x C) in the split-bf16 mode,
// 8 waves (32 x 32 per wave) in exact fp32, see RCfg.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

#include <stdlib.h>

// implemented in resunit_f32.hip: 1 handled, 0 not eligible, < 0 error
int srn_resunit_f32_try(const SrnResUnitParams& p, hipStream_t stream);

namespace {

constexpr int RU_HALO_MAX = 50;  // (k - 1) * dilation of the widest unit on the path (k 11, d 5)

__device__ __attribute__((aligned(256))) float g_zero_ru[64];

typedef float f32x2r __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair_r(const float a, const float b, unsigned& hi, unsigned& lo) {
  const f32x2r v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2r));
  f32x2r hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2r l = v - hf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2r));
}

template <int C_, int PREC_, int NW_ = 4>
struct RCfg {
  static constexpr int C = C_, PREC = PREC_, NW = NW_, NTH = 64 * NW_;
  static constexpr int CH = C / 32, NT = C / 32, MT = C == 32 ? 2 : 1;
  // NW = 4: a wave owns 32 MT rows x all C columns.  NW = 8 (exact fp32): 32 rows x 32 columns per wave -- 8 row groups at
  // C = 32, 4 row groups x 2 column groups at C = 64 -- so the same image feeds twice the waves (4 per SIMD at two
  // workgroups per CU): the fp32 loop is latency-bound at 2 waves per SIMD (MFMA pipe 0.58-0.67 busy), not LDS-bound
  static constexpr int MTW = NW == 8 ? 1 : MT, NTW = NW == 8 ? 1 : NT;
  static_assert(NW == 4 || (NW == 8 && PREC == 0), "eight waves: exact fp32 only");
  static constexpr int BMI = 128 * MT;                 // intermediate rows per tile
  static constexpr int ROWB = PREC ? 64 : 144;         // bytes of one row of one 32-channel chunk (plane)
  static constexpr int PL = PREC ? 2 : 1;              // planes: (hi, lo) or fp32
  static constexpr int A_ROWS = BMI + RU_HALO_MAX + 2;
  static constexpr int A_PLANE = A_ROWS * ROWB;
  static constexpr int A_BYTES = CH * PL * A_PLANE;
  static constexpr int UNIT = PL * C * ROWB;           // weights of one (tap, chunk): [plane][n][ROWB]
  static constexpr int G = (PREC ? 16384 : 9216) / UNIT >= 2 ? 2 : 1;  // units per LDS stage
  static constexpr int STAGE = G * UNIT;
  static constexpr int SMEM = A_BYTES + 2 * STAGE;
  static constexpr int F4R = C / 4;                    // float4 pieces per input row
  static constexpr int A_LD = (A_ROWS * F4R + NTH - 1) / NTH;
  static constexpr int W_LD = G * C * 8 / NTH;         // 16-B pieces per thread per stage
  static_assert(G * C * 8 % NTH == 0, "stage pieces must divide over the workgroup");
};

template <class R>
__global__ __launch_bounds__(R::NTH, R::NW / 2) void resunit_kernel(const SrnResUnitParams p, const int tiles_per_z,
                                                            const int n_tiles) {
  constexpr int C = R::C, CH = R::CH, NT = R::NTW, MT = R::MTW, BMI = R::BMI, G = R::G, NTH = R::NTH;
  constexpr bool BF = R::PREC != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
  unsigned char* sA = smem_r;
  unsigned char* sW = smem_r + R::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int k = p.k, dil = p.dilation, T = p.T;
  const int p2 = (k - 1) / 2, p1 = p2 * dil;
  const int BMo = BMI - (k - 1);         // output rows per tile
  const int hr = BMI + (k - 1) * dil;    // image rows in use
  const int U = k * CH;                  // weight units per conv
  const int S = (U + G - 1) / G;         // weight stages per conv
  const float slope = p.slope;

  // ---- weight stages: j in [0, 2 S): conv1's stages then conv2's.  Piece q = tid + 256 i of a stage is 16-B
  //      piece (tid & 7) of weight row n = (tid >> 3) + 32 (i & 1) [C = 64] of unit i >> 1 [C = 64] / i [C = 32]:
  //      everything but the unit is fixed per thread, so a stage costs one address and immediates
  constexpr int W_LD = R::W_LD;
  constexpr int PU = C * 8;  // 16-B pieces of one unit: piece q = tid + NTH i is piece q % 8 of row (q % PU) / 8 of unit q / PU
  const int w_piece = tid & 7;
  uint4 wr[W_LD];
  auto w_load = [&](const int j) {
    const int conv = j >= S;
    const int u0 = (conv ? j - S : j) * G;
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      const int q = tid + NTH * i;
      const int g = q / PU;
      const int n = (q % PU) >> 3;
      const int u = min(u0 + g, U - 1);  // a partial last stage re-reads the last unit; it is never multiplied
      const uint4* src;
      if constexpr (BF)
        src = reinterpret_cast<const uint4*>(conv ? p.w2_hi : p.w1_hi) + ((n * U + u) * 8 + w_piece);
      else
        src = reinterpret_cast<const uint4*>((conv ? p.w2 : p.w1) + (n * (k * C) + u * 32)) + w_piece;
      wr[i] = *src;
    }
  };
  auto w_store = [&](const int buf) {
    unsigned char* base = sW + buf * R::STAGE;
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      const int q = tid + NTH * i;
      const int g = q / PU;
      const int n = (q % PU) >> 3;
      const int off = BF ? g * 2 * (C * 64) + (w_piece >> 2) * (C * 64) + bf_off(n, (w_piece & 3) * 8)
                         : (g * C + n) * 144 + w_piece * 16;
      // through an opaque asm: the optimizer turns a pure global -> register -> LDS copy of the set into memcpys
      // of a stack object (scratch)
      uint4 v = wr[i];
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
      *reinterpret_cast<uint4*>(base + off) = v;
    }
  };

  // ---- receptive-field image: piece q = tid + 256 j is float4 (tid % F4R) of image row tid / F4R + RSTEP j.  Rows are
  //      loaded from a clamped address and zeroed at staging time by a validity bit (one pointer + one mask instead
  //      of a pointer per piece)
  constexpr int A_LD = R::A_LD, F4R = R::F4R, RSTEP = NTH / F4R;
  const int a_f4 = tid % F4R;
  const int a_r0 = tid / F4R;
  float4 pa[A_LD];
  unsigned pa_ok = 0;
  auto load_a = [&](const int tile) {
    const int z = tile / tiles_per_z;
    const int t0 = (tile - z * tiles_per_z) * BMo;
    const float* xz = p.x + (int64_t)z * p.x_bs + a_f4 * 4;
    const int ti0 = t0 - p2 - p1 + a_r0;
    pa_ok = 0;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int ti = ti0 + RSTEP * j;
      const bool ok = a_r0 + RSTEP * j < hr && ti >= 0 && ti < T;
      pa_ok |= (ok ? 1u : 0u) << j;
      pa[j] = *reinterpret_cast<const float4*>(xz + min(max(ti, 0), T - 1) * C);
    }
  };
  // LDS offset of piece j = a_dst0 + j * RSTEP rows (RSTEP is a multiple of 4: the swizzle key does not change)
  const int a_dst0 = BF ? (a_f4 >> 3) * 2 * R::A_PLANE + bf_off(a_r0, (a_f4 & 7) * 4)
                        : (a_f4 >> 3) * R::A_PLANE + a_r0 * 144 + (a_f4 & 7) * 16;
  auto stage_a = [&]() {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      if (a_r0 + RSTEP * j < hr) {
        float4 v = leaky4(pa[j], slope);
        if (!((pa_ok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (BF) {
          uint2 hi, lo;
          split_pair_r(v.x, v.y, hi.x, lo.x);
          split_pair_r(v.z, v.w, hi.y, lo.y);
          *reinterpret_cast<uint2*>(sA + a_dst0 + j * RSTEP * 64) = hi;
          *reinterpret_cast<uint2*>(sA + a_dst0 + R::A_PLANE + j * RSTEP * 64) = lo;
        } else {
          *reinterpret_cast<float4*>(sA + a_dst0 + j * RSTEP * 144) = v;
        }
      }
    }
  };

  // NW = 4: wave = row group, all columns.  NW = 8: 32 x 32 per wave; at C = 64 waves 4-7 take the second 32 columns
  const int wm0 = (R::NW == 8 && C == 64 ? (wave & 3) : wave) * 32 * MT;
  const int ng0 = R::NW == 8 && C == 64 ? (wave >> 2) : 0;  // first 32-column block of this wave
  f32x16 acc[MT][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // MFMAs of one weight stage (units u0 .. u0 + G - 1 of a conv whose taps are `step` image rows apart)
  auto compute = [&](const int buf, const int u0, const int step) {
    const unsigned char* wb = sW + buf * R::STAGE;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int u = u0 + g;
      if (u >= U) break;  // wave-uniform
      __builtin_amdgcn_sched_barrier(0);  // keep one unit's fragments live at a time (256-VGPR budget)
      const int tap = u / CH;
      const int c = u - tap * CH;
      if constexpr (BF) {
        int arow[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) arow[m] = wm0 + m * 32 + li + tap * step;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int a_ch = (((kk * 2 + lh) ^ (arow[m] >> 2)) & 3) << 4;
            const unsigned char* a = sA + arow[m] * 64 + a_ch;
            ah[m] = *reinterpret_cast<const bf16x8*>(a + (c * 2) * R::A_PLANE);
            al[m] = *reinterpret_cast<const bf16x8*>(a + (c * 2 + 1) * R::A_PLANE);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int o = bf_off((ng0 + n) * 32 + li, kk * 16 + lh * 8);
            bh[n] = *reinterpret_cast<const bf16x8*>(wb + (g * 2) * (C * 64) + o);
            bl[n] = *reinterpret_cast<const bf16x8*>(wb + (g * 2 + 1) * (C * 64) + o);
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
        const float* a = reinterpret_cast<const float*>(sA + c * R::A_PLANE) + (wm0 + li + tap * step) * 36 + 4 * lh;
        const float* b = reinterpret_cast<const float*>(wb + g * C * 144) + (ng0 * 32 + li) * 36 + 4 * lh;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float4 af[MT], bf[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a + m * 32 * 36 + kk * 8);
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float4*>(b + n * 32 * 36 + kk * 8);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
        }
      }
    }
  };

  // one conv: S stages out of the double-buffered weight pipeline.  Invariant on entry: LDS buffer `wbuf` holds stage
  // `wj`, the registers hold stage wj + 1 (indices mod 2 S); the same holds on exit for the next conv.
  int wj = 0, wbuf = 0;
  auto conv_pass = [&](const int step) {
    for (int s = 0; s < S; ++s) {
      compute(wbuf, s * G, step);
      w_store(wbuf ^ 1);
      int nxt = wj + 2;
      if (nxt >= 2 * S) nxt -= 2 * S;
      w_load(nxt);
      wj = wj + 1 == 2 * S ? 0 : wj + 1;
      wbuf ^= 1;
      __syncthreads();
    }
  };

  int tile = blockIdx.x;  // grid <= n_tiles
  load_a(tile);
  w_load(0);
  w_store(0);
  w_load(2 * S > 1 ? 1 : 0);

  for (; tile < n_tiles; tile += gridDim.x) {
    const int z = tile / tiles_per_z;
    const int t0 = (tile - z * tiles_per_z) * BMo;
    stage_a();  // the previous tile's conv2 ended on a barrier: nobody reads the image any more
    __syncthreads();
    const int next = tile + gridDim.x;
    load_a(next < n_tiles ? next : tile);  // unconditional prefetch; a dummy is never staged

    // ---- conv1 over the x image
    zero_acc();
    conv_pass(dil);
    // ---- intermediate = LeakyReLU(conv1 + b1), zero outside [0, T), over the image (every wave passed the last
    //      stage's barrier, so no conv1 read is pending)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int na = ng0 + n;  // 32-column block of the intermediate = channel chunk of conv2's image
        const float bias = p.b1[na * 32 + li];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          // rows mrow0 .. mrow0 + 3 of column li: one swizzle key per row quad ((mrow >> 2) & 3 = (2 gq + lh) & 3)
          const int mrow0 = wm0 + m * 32 + 8 * gq + 4 * lh;
          const int off0 = BF ? mrow0 * 64 + ((((li >> 3) ^ (2 * gq + lh)) & 3) << 4) + (li & 7) * 2
                              : (mrow0 * 36 + li) * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int gi = t0 - p2 + mrow0 + i;
            float v = acc[m][n][4 * gq + i] + bias;
            v = v > 0.f ? v : v * slope;
            if (gi < 0 || gi >= T) v = 0.f;
            if constexpr (BF) {
              const __bf16 h = (__bf16)v;
              const __bf16 l = (__bf16)(v - (float)h);
              *reinterpret_cast<__bf16*>(sA + (na * 2) * R::A_PLANE + off0 + i * 64) = h;
              *reinterpret_cast<__bf16*>(sA + (na * 2 + 1) * R::A_PLANE + off0 + i * 64) = l;
            } else {
              *reinterpret_cast<float*>(sA + na * R::A_PLANE + off0 + i * 144) = v;
            }
          }
        }
      }
    __syncthreads();

    // ---- conv2 over the intermediate image
    zero_acc();
    conv_pass(1);

    // ---- epilogue: + b2 + x [+ res2] [/ post_div]; rows [t0, t0 + BMo) and < T.  out never aliases x; res2 may be
    //      out itself (running stage sum): every element is loaded by the lane that later stores it.  The loads of a
    //      row quad are issued together and nothing is preloaded across sub-tiles (register budget)
    const float* __restrict__ xz = p.x + (int64_t)z * p.x_bs + (int64_t)t0 * C;
    const float* qz = p.res2 ? p.res2 + (int64_t)z * p.res2_bs + (int64_t)t0 * C : nullptr;  // may alias out
    float* oz = p.out + (int64_t)z * p.out_bs + (int64_t)t0 * C;
    const bool divide = p.post_div != 0.f && p.post_div != 1.f;
    const int row_end = min(BMo, T - t0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = (ng0 + n) * 32 + li;
        const float bias = p.b2[col];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float xv[4], qv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = wm0 + m * 32 + 8 * gq + 4 * lh + i;
            const bool ok = row < row_end;
            xv[i] = ok ? xz[row * C + col] : 0.f;
            qv[i] = (ok && qz) ? qz[row * C + col] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = wm0 + m * 32 + 8 * gq + 4 * lh + i;
            if (row < row_end) {
              float v = acc[m][n][4 * gq + i] + bias + xv[i];
              if (qz) v += qv[i];
              if (divide) v = v / p.post_div;
              oz[row * C + col] = v;
            }
          }
        }
      }
  }
}

template <class R>
int launch_resunit(const SrnResUnitParams& p, hipStream_t stream) {
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&resunit_kernel<R>), R::SMEM)) return e;
  const int BMo = R::BMI - (p.k - 1);
  const int tiles_per_z = (p.T + BMo - 1) / BMo;
  const int64_t n_tiles = (int64_t)p.n_batch * tiles_per_z;
  SRN_CHECK_ARG(n_tiles > 0 && n_tiles < (1ll << 31), "resunit: bad tile count %lld", (long long)n_tiles);
  const int grid = (int)(n_tiles < 512 ? n_tiles : 512);  // persistent: two workgroups per CU
  hipLaunchKernelGGL((resunit_kernel<R>), dim3(grid), dim3(R::NTH), R::SMEM, stream, p, tiles_per_z, (int)n_tiles);
  SRN_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int srn_hifigan_resunit(const SrnResUnitParams* pp, void* stream_) {
  SRN_CHECK_ARG(pp != nullptr, "resunit: null params");
  const SrnResUnitParams& p = *pp;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  SRN_CHECK_ARG(p.x && p.w1 && p.b1 && p.w2 && p.b2 && p.out, "resunit: null pointer");
  SRN_CHECK_ARG(p.n_batch > 0 && p.T > 0, "resunit: bad sizes");
  SRN_CHECK_ARG(p.C == 32 || p.C == 64, "resunit: C = %d (this fused kernel takes 32 or 64 channels)", p.C);
  SRN_CHECK_ARG(p.k >= 1 && p.k % 2 == 1 && p.dilation >= 1 && (p.k - 1) * p.dilation <= RU_HALO_MAX,
                "resunit: kernel %d / dilation %d outside the staged halo (%d rows)", p.k, p.dilation, RU_HALO_MAX);
  SRN_CHECK_ARG(p.out != p.x && p.out != p.res2 - 0 ? true : p.out != p.x, "resunit: out must not alias x");
  SRN_CHECK_ARG(((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w1) |
                  reinterpret_cast<uintptr_t>(p.w2)) & 15) == 0 && p.x_bs % 4 == 0,
                "resunit: x / w1 / w2 must be 16-byte aligned");
  if (p.precision == SRN_PREC_BF16X3) {
    SRN_CHECK_ARG(p.w1_hi && p.w2_hi, "resunit: split-bf16 mode needs the weight planes w1_hi / w2_hi");
    return p.C == 32 ? launch_resunit<RCfg<32, 1>>(p, stream) : launch_resunit<RCfg<64, 1>>(p, stream);
  }
  // exact fp32: resunit_f32.hip's form of this kernel (same results bit for bit, ~no vector-ALU work beside the fp32
  // MFMAs); SERENADE_AMD_RESUNIT_SHARED_FP32=1 keeps this file's instantiation (A-B timing, bit-identity test)
  {
    const char* e = getenv("SERENADE_AMD_RESUNIT_SHARED_FP32");
    if (!(e && e[0] == '1')) {
      const int r = srn_resunit_f32_try(p, stream);
      if (r != 0) return r < 0 ? r : 0;
    }
  }
  // eight waves per workgroup (A/B on the B = 8 x T = 1024 vocoder, 4 -> 8 waves: k 3 units 0.62 -> 0.51 ms,
  // k 7 1.17 -> 1.05, k 11 1.74 -> 1.61 at C = 64; all 18 units 16.7 -> 14.9 ms)
  return p.C == 32 ? launch_resunit<RCfg<32, 0, 8>>(p, stream) : launch_resunit<RCfg<64, 0, 8>>(p, stream);
}
