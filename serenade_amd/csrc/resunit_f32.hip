// resunit_f32.hip -- the exact-fp32 form of the fused HiFi-GAN residual unit (resunit.hip: same algorithm, LDS image,
// weight pipeline, MFMA order and results, bit for bit), rewritten for the fp32 matrix pipe's economics.
//
// v_mfma_f32_32x32x2_f32 shares the SIMD's fp32 lanes with ordinary vector instructions (conv_f32.hip, round 4): every
// VALU instruction a wave issues is 3-7 cycles the matrix stream does not get.  The shared kernel's fp32 instantiation
// issued ~15 of them per 16-MFMA weight stage (64-bit weight addresses, a quarter-rate integer multiply for the image
// row, clamps) and ~35 per staged float4 of the receptive field (clamped 64-bit addresses, compare + select LeakyReLU,
// a zero select): 10-20 % of the matrix time, most at k = 3.  Here
//   * weights, the receptive field, the exact residual and the output go through buffer instructions: wave-uniform
//     descriptors, per-lane 32-bit offsets computed once (weights) or with one add per tile (rows), the weight walk in
//     the SCALAR offset; rows outside [0, T) fall outside the descriptor and read as zeros (no clamp, no select);
//   * fragment addresses are one scalar-plus-vector add per (tap, chunk) unit;
//   * LeakyReLU is max(x, slope x) (two instructions; 0 <= slope <= 1 host-checked);
//   * interior tiles take paths without per-element bounds code.
// Reference: HiFiGANResidualBlock.forward, serenade/vocoder/layers/residual_block.py:243-258; stage sum / mean of
// hifigan.py:183-186 in the epilogue.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int HALO_MAX = 50;  // (k - 1) * dilation of the widest unit on the path (k 11, d 5): resunit.hip's RU_HALO_MAX
constexpr int NTH = 512;      // eight waves: 32 x 32 of the tile per wave (resunit.hip, RCfg<C, 0, 8>)

typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

struct FDivR {
  uint32_t mul, shift;
};
__device__ __forceinline__ int fdivr(const int n, const FDivR d) {
  return (int)(((uint64_t)(uint32_t)n * d.mul) >> d.shift);
}

template <int C_>
struct UCfg {
  static constexpr int C = C_, CH = C / 32;
  static constexpr int BMI = C == 32 ? 256 : 128;  // intermediate rows per tile (8 or 4 row groups of 32)
  static constexpr int A_ROWS = BMI + HALO_MAX + 2;
  static constexpr int A_PLANE = A_ROWS * 144;     // one 32-channel chunk of the image: 144-B rows
  static constexpr int A_BYTES = CH * A_PLANE;
  static constexpr int UNIT = C * 144;             // weights of one (tap, chunk): [n][144 B]
  static constexpr int G = 9216 / UNIT >= 2 ? 2 : 1;
  static constexpr int STAGE = G * UNIT;
  static constexpr int SMEM = A_BYTES + 2 * STAGE;
  static constexpr int F4R = C / 4, RSTEP = NTH / F4R;
  static constexpr int A_LD = (A_ROWS * F4R + NTH - 1) / NTH;
  static_assert(G * C * 8 == NTH, "one 16-B weight piece per thread per stage");
};

template <class R>
__global__ __launch_bounds__(NTH, 4) void resunit_f32_kernel(const SrnResUnitParams p, const int tiles_per_z,
                                                             const int n_tiles, const FDivR d_tpz) {
  constexpr int C = R::C, CH = R::CH, BMI = R::BMI, G = R::G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];
  unsigned char* const sA = smem_u;
  unsigned char* const sW = smem_u + R::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int k = p.k, dil = p.dilation, T = p.T;
  const int p2 = (k - 1) / 2, p1 = p2 * dil;
  const int BMo = BMI - (k - 1);       // output rows per tile
  const int hr = BMI + (k - 1) * dil;  // image rows in use
  const int U = k * CH;                // weight units per conv
  const int S = (U + G - 1) / G;       // weight stages per conv
  const float slope = p.slope;

  // ---- weight stages j in [0, 2 S): conv1's then conv2's.  This thread's piece: 16-B piece (tid & 7) of weight row n
  //      of unit u0 + g of the stage -- row and piece in the per-lane offset, the unit in the scalar offset
  const int w_n = (tid % (C * 8)) >> 3;
  const int w_g = tid / (C * 8);
  const int w_voff = w_n * (k * C * 4) + w_g * 128 + (tid & 7) * 16;
  const int w_dst = (w_g * C + w_n) * 144 + (tid & 7) * 16;
  const int w_bytes = C * k * C * 4;
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, w_bytes, 0x00020000);
  u32x4r wr;
  auto w_load = [&](const int j) {  // a partial last stage reads on into the row (or past the tensor: zeros); never multiplied
    if (j >= S) wr = __builtin_amdgcn_raw_buffer_load_b128(rs_w2, w_voff, (j - S) * G * 128, 0);
    else wr = __builtin_amdgcn_raw_buffer_load_b128(rs_w1, w_voff, j * G * 128, 0);
  };
  auto w_store = [&](const int buf) {
    *reinterpret_cast<u32x4r*>(sW + buf * R::STAGE + w_dst) = wr;
  };

  // ---- receptive-field image: piece j of this thread is float4 (tid % F4R) of image row tid / F4R + RSTEP j
  constexpr int A_LD = R::A_LD, F4R = R::F4R, RSTEP = R::RSTEP;
  const int a_f4 = tid % F4R;
  const int a_r0 = tid / F4R;
  const int a_voff = (a_r0 * C + a_f4 * 4) * 4;
  const int a_dst0 = (a_f4 >> 3) * R::A_PLANE + a_r0 * 144 + (a_f4 & 7) * 16;
  u32x4r pa[A_LD];
  auto load_a = [&](const int tile) {
    const int z = fdivr(tile, d_tpz);
    const int t0 = (tile - z * tiles_per_z) * BMo;
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)z * p.x_bs), 0, T * C * 4, 0x00020000);
    const int v0 = a_voff + (t0 - p2 - p1) * (C * 4);  // negative rows wrap to offsets >= 2^31: out of range, zeros
#pragma unroll
    for (int j = 0; j < A_LD; ++j) pa[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, v0 + j * (RSTEP * C * 4), 0, 0);
  };
  auto stage_a = [&]() {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      if (a_r0 + RSTEP * j < hr) {
        float4 v = __builtin_bit_cast(float4, pa[j]);
        v.x = fmaxf(v.x, v.x * slope);
        v.y = fmaxf(v.y, v.y * slope);
        v.z = fmaxf(v.z, v.z * slope);
        v.w = fmaxf(v.w, v.w * slope);
        *reinterpret_cast<float4*>(sA + a_dst0 + j * RSTEP * 144) = v;
      }
    }
  };

  // 32 x 32 per wave; at C = 64 waves 4-7 take the second 32 columns
  const int wm0 = (C == 64 ? (wave & 3) : wave) * 32;
  const int ng0 = C == 64 ? (wave >> 2) : 0;  // 32-column block of this wave
  const int fa_lane = (wm0 + li) * 144 + 16 * lh;                     // A fragments: byte offset inside a plane
  const int fb_lane = R::A_BYTES + (ng0 * 32 + li) * 144 + 16 * lh;   // B fragments: inside stage 0, unit 0
  f32x16 acc;
  auto zero_acc = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };

  // MFMAs of one weight stage (units u0 .. u0 + G - 1 of a conv whose taps are `step` image rows apart)
  auto compute = [&](const int buf, const int u0, const int step) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int u = u0 + g;
      if (u >= U) break;  // wave-uniform
      const int tap = u / CH;
      const int c = u - tap * CH;
      const float* a = reinterpret_cast<const float*>(smem_u + (fa_lane + (c * R::A_PLANE + tap * step * 144)));
      const float* b = reinterpret_cast<const float*>(smem_u + (fb_lane + (buf * R::STAGE + g * C * 144)));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float4 af = *reinterpret_cast<const float4*>(a + kk * 8);
        const float4 bf = *reinterpret_cast<const float4*>(b + kk * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
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

  const float bias1 = p.b1[ng0 * 32 + li];
  const float bias2 = p.b2[ng0 * 32 + li];
  const bool divide = p.post_div != 0.f && p.post_div != 1.f;
  const bool has_q = p.res2 != nullptr;
  const int mid_lane = ng0 * R::A_PLANE + ((wm0 + 4 * lh) * 36 + li) * 4;  // intermediate element (row wm0 + 4 lh, col li)
  const int io_lane = ((wm0 + 4 * lh) * C + ng0 * 32 + li) * 4;          // x / res2 / out element of that row, tile-relative

  for (; tile < n_tiles; tile += gridDim.x) {
    const int z = fdivr(tile, d_tpz);
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
    if (t0 - p2 >= 0 && t0 - p2 + BMI <= T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[r] + bias1;
        *reinterpret_cast<float*>(sA + mid_lane + ((r & 3) + 8 * (r >> 2)) * 144) = fmaxf(v, v * slope);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const int gi = t0 - p2 + wm0 + 4 * lh + dr;
        float v = acc[r] + bias1;
        v = fmaxf(v, v * slope);
        if (gi < 0 || gi >= T) v = 0.f;
        *reinterpret_cast<float*>(sA + mid_lane + dr * 144) = v;
      }
    }
    __syncthreads();

    // ---- conv2 over the intermediate image
    zero_acc();
    conv_pass(1);

    // ---- epilogue: + b2 + x [+ res2] [/ post_div]; rows [t0, t0 + BMo) and < T.  out never aliases x; res2 may be
    //      out itself (running stage sum): every element is loaded by the lane that later stores it
    const int row_end = min(BMo, T - t0);
    const int64_t zoff = (int64_t)t0 * C;
    const int io_bytes = row_end * C * 4;  // the tile's rows that exist: anything past them is out of range
    const __amdgpu_buffer_rsrc_t rs_xe = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)z * p.x_bs + zoff), 0, io_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_qe = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_q ? p.res2 + (int64_t)z * p.res2_bs + zoff : p.x), 0, has_q ? io_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_oe =
        __builtin_amdgcn_make_buffer_rsrc(p.out + (int64_t)z * p.out_bs + zoff, 0, io_bytes, 0x00020000);
    if (wm0 + 32 <= row_end) {  // every row of this wave exists: row in the scalar offset
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float xv[4], qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xe, io_lane, (8 * gq + i) * C * 4, 0));
          qv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_qe, io_lane, (8 * gq + i) * C * 4, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[4 * gq + i] + bias2 + xv[i];
          if (has_q) v += qv[i];
          if (divide) v = v / p.post_div;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_oe, io_lane, (8 * gq + i) * C * 4, 0);
        }
      }
    } else if (wm0 < row_end) {  // the tile's last rows: the row in the per-lane offset, so the range check sees it
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float xv[4], qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_xe, io_lane + (8 * gq + i) * C * 4, 0, 0));
          qv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_qe, io_lane + (8 * gq + i) * C * 4, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[4 * gq + i] + bias2 + xv[i];
          if (has_q) v += qv[i];
          if (divide) v = v / p.post_div;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_oe, io_lane + (8 * gq + i) * C * 4, 0, 0);
        }
      }
    }
  }
}

FDivR make_fdivr(const uint32_t d) {
  int lg = 0;
  while ((1u << lg) < d) ++lg;
  const int k = 26 + lg;
  return FDivR{(uint32_t)(((1ull << k) + d - 1) / d), (uint32_t)k};
}

template <class R>
int launch_unit(const SrnResUnitParams& p, hipStream_t stream) {
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&resunit_f32_kernel<R>), R::SMEM)) return e;
  const int BMo = R::BMI - (p.k - 1);
  const int tiles_per_z = (p.T + BMo - 1) / BMo;
  const int64_t n_tiles = (int64_t)p.n_batch * tiles_per_z;
  if (n_tiles <= 0 || n_tiles >= (1ll << 26)) return 0;
  const int grid = (int)(n_tiles < 512 ? n_tiles : 512);  // persistent: two workgroups per CU
  hipLaunchKernelGGL((resunit_f32_kernel<R>), dim3(grid), dim3(NTH), R::SMEM, stream, p, tiles_per_z, (int)n_tiles,
                     make_fdivr((uint32_t)tiles_per_z));
  SRN_CHECK_LAUNCH();
  return 1;
}

}  // namespace

// Returns 1 if the launch was handled, 0 if the unit is not eligible (the caller runs resunit.hip's fp32 form), < 0 on
// error.  `p` has been validated by srn_hifigan_resunit.
int srn_resunit_f32_try(const SrnResUnitParams& p, hipStream_t stream) {
  if (!(p.slope >= 0.f && p.slope <= 1.f)) return 0;
  if ((int64_t)p.T * p.C * 4 >= 0x7fffffffll) return 0;  // 32-bit byte offsets inside one item
  return p.C == 32 ? launch_unit<UCfg<32>>(p, stream) : launch_unit<UCfg<64>>(p, stream);
}
