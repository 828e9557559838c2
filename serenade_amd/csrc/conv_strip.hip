// conv_strip.hip -- thin split-bf16 convs: C_in and N in {32, 64}, stride 1, static weights (the last two HiFi-GAN
// stages, HiFiGANResidualBlock `residual_block.py:187-258` at 64 / 32 channels, and the SiFiGAN blocks of that width).
//
// On the tiled kernels these shapes give 6-12 MFMAs per wave between two barriers and re-stage the weight tile per
// tap; measured 50-110 TFLOP/s, 2-3x above both their HBM and their MFMA bound.  Here a persistent workgroup keeps
// the WHOLE weight tensor of the conv in LDS as pre-split bf16 hi|lo planes (<= 115 KB) and walks 128-row output
// tiles: per tile it stages the receptive-field image (128 + halo rows, split once) and runs every tap out of LDS --
// 66-132 MFMAs per wave between barriers, no per-tap staging.  The next tile's rows are prefetched into registers
// under the MFMAs.  Per output row the kernel moves the algorithmic bytes only: C_in*4 in, N*4 out (+ residuals).
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int BM = 128;        // output rows per tile: 4 waves x 32 rows
constexpr int HALO_MAX = 52;   // (k - 1) * dilation of the widest conv on the path (k 11, d 5 -> 50)
constexpr int HR_MAX = BM + HALO_MAX;

__device__ __attribute__((aligned(256))) float g_zero_strip[64];

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_pair_s(const float a, const float b, unsigned& hi, unsigned& lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 hf;
  hf.x = __builtin_bit_cast(float, hi << 16);
  hf.y = __builtin_bit_cast(float, hi & 0xffff0000u);
  const f32x2 l = v - hf;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
}

// CIN: input channels (32 / 64); NT: N / 32 (1 / 2); ACT: SRN_ACT_NONE / SRN_ACT_LEAKY.
// LDS: W [tap][chunk][hi|lo][N rows][64 B swizzled] then A [chunk][hi|lo][HR_MAX rows][64 B swizzled].
template <int CIN, int NT, int ACT>
__global__ __launch_bounds__(256, 2) void conv_strip_kernel(const SrnConvParams p, const int min_off, const int halo,
                                                            const int tiles_per_z, const int n_tiles) {
  constexpr int CH = CIN / 32;         // 32-channel chunks
  constexpr int N = NT * 32;
  constexpr int F4R = CIN / 4;         // float4 pieces per input row
  constexpr int A_LD = (HR_MAX * F4R + 255) / 256;
  constexpr int A_PLANE = HR_MAX * 64;  // bytes of one (chunk, plane) image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int n_taps = p.n_taps;
  unsigned char* sW = smem_s;
  unsigned char* sA = smem_s + n_taps * CH * 2 * N * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int hr = BM + halo;  // image rows actually used

  // ---- weights: once per workgroup, 16-B pieces of the plane image [N][tap][chunk][hi 64 B | lo 64 B]
  {
    const int per_n = n_taps * CH * 8;
    const int pieces = N * per_n;
    const uint4* src = reinterpret_cast<const uint4*>(p.w_hi);
    for (int q = tid; q < pieces; q += 256) {
      const int n = q / per_n;
      const int r = q - n * per_n;
      const int tc = r >> 3;  // tap * CH + chunk
      const int piece = r & 7;
      const uint4 v = src[q];
      *reinterpret_cast<uint4*>(sW + ((tc * 2 + (piece >> 2)) * N) * 64 + bf_off(n, (piece & 3) * 8)) = v;
    }
  }

  // ---- per-thread pieces of the A image: piece q = tid + 256 j -> (row q / F4R, float4 q % F4R)
  float4 pa[A_LD];
  auto load_a = [&](const int tile) {
    const int z = tile / tiles_per_z;
    const int t0 = (tile - z * tiles_per_z) * BM;
    const float* in0 = p.in0 + (int64_t)z * p.in0_bs;
    int len_in = p.T_in;
    if (p.len_in) len_in = min(p.len_in[z], p.T_in);
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int q = tid + 256 * j;
      const int row = q / F4R;
      const int f4 = q - row * F4R;
      const int ti = t0 + min_off + row;
      const bool ok = row < hr && ti >= 0 && ti < len_in;
      const float* src = ok ? in0 + (int64_t)ti * p.ld_in0 + f4 * 4 : g_zero_strip + (f4 & 7) * 4;
      pa[j] = *reinterpret_cast<const float4*>(src);
    }
  };
  const float pro_slope = p.pro_slope;
  auto stage_a = [&]() {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int q = tid + 256 * j;
      const int row = q / F4R;
      const int f4 = q - row * F4R;
      if (row < hr) {
        float4 v = pa[j];
        if constexpr (ACT == SRN_ACT_LEAKY) {
          v.x = v.x > 0.f ? v.x : v.x * pro_slope;
          v.y = v.y > 0.f ? v.y : v.y * pro_slope;
          v.z = v.z > 0.f ? v.z : v.z * pro_slope;
          v.w = v.w > 0.f ? v.w : v.w * pro_slope;
        }
        uint2 hi, lo;
        split_pair_s(v.x, v.y, hi.x, lo.x);
        split_pair_s(v.z, v.w, hi.y, lo.y);
        const int chunk = f4 >> 3;
        const int off = bf_off(row, (f4 & 7) * 4);
        *reinterpret_cast<uint2*>(sA + (chunk * 2) * A_PLANE + off) = hi;
        *reinterpret_cast<uint2*>(sA + (chunk * 2 + 1) * A_PLANE + off) = lo;
      }
    }
  };

  const int wm0 = wave * 32;
  int tile = blockIdx.x;
  load_a(tile);  // grid <= n_tiles
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();  // every wave is done reading the previous image (first pass: nothing)
    stage_a();
    __syncthreads();  // image (and, first pass, the weights) visible
    const int next = tile + gridDim.x;
    load_a(next < n_tiles ? next : tile);  // unconditional prefetch (keeps vmcnt counted); the dummy is not staged

    f32x16 acc[1][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][n][r] = 0.f;

    for (int tap = 0; tap < n_taps; ++tap) {
      const int arow = wm0 + li + (p.tap_off[tap] - min_off);  // image row of this lane's output row at this tap
      const int asw = (arow >> 2) & 3;
      const unsigned char* a_base = sA + arow * 64;
      const unsigned char* w_base = sW + (tap * CH * 2 * N) * 64;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int a_ch = (((kk * 2 + lh) ^ asw) & 3) << 4;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_base + (c * 2) * A_PLANE + a_ch);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(a_base + (c * 2 + 1) * A_PLANE + a_ch);
          bf16x8 bh[NT], bl[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int o = bf_off(n * 32 + li, kk * 16 + lh * 8);
            bh[n] = *reinterpret_cast<const bf16x8*>(w_base + (c * 2) * N * 64 + o);
            bl[n] = *reinterpret_cast<const bf16x8*>(w_base + (c * 2 + 1) * N * 64 + o);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[n], acc[0][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[n], acc[0][n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[n], acc[0][n], 0, 0, 0);
        }
      }
    }

    const int z = tile / tiles_per_z;
    const int t0 = (tile - z * tiles_per_z) * BM;
    conv_epilogue<1, NT>(p, acc, z, 0, t0, 0, wm0, 0, lane);
  }
}

template <int CIN, int NT, int ACT>
int launch_strip(const SrnConvParams& p, int min_off, int halo, hipStream_t stream) {
  const int smem = p.n_taps * (CIN / 32) * 2 * (NT * 32) * 64 + (CIN / 32) * 2 * HR_MAX * 64;
  static SrnSmemAttr smem_attr;  // granted once per device at the most this kernel ever asks for
  SRN_CHECK_ARG(smem <= 160 * 1024, "conv_strip: %d B of LDS", smem);
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_strip_kernel<CIN, NT, ACT>), 160 * 1024)) return e;
  const int tiles_per_z = (p.T_out + BM - 1) / BM;
  const int64_t n_tiles = (int64_t)p.n_batch * tiles_per_z;
  SRN_CHECK_ARG(n_tiles > 0 && n_tiles < (1ll << 31), "conv_strip: bad tile count %lld", (long long)n_tiles);
  int per_cu = (156 * 1024) / smem;
  per_cu = per_cu < 1 ? 1 : per_cu;  // workgroups that fit a CU's 160 KB (register use allows up to 4 waves per SIMD for C_in 32)
  per_cu = per_cu > 4 ? 4 : per_cu;
  const int grid = (int)(n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu);
  hipLaunchKernelGGL((conv_strip_kernel<CIN, NT, ACT>), dim3(grid), dim3(256), smem, stream, p, min_off, halo,
                     tiles_per_z, (int)n_tiles);
  SRN_CHECK_LAUNCH();
  return 1;
}

template <int CIN, int NT>
int launch_strip_act(const SrnConvParams& p, int min_off, int halo, hipStream_t stream) {
  if (p.pro_act == SRN_ACT_LEAKY) return launch_strip<CIN, NT, SRN_ACT_LEAKY>(p, min_off, halo, stream);
  return launch_strip<CIN, NT, SRN_ACT_NONE>(p, min_off, halo, stream);
}

}  // namespace

// Returns 1 if the launch was handled, 0 if the shape is not eligible, < 0 on error.
int srn_conv_strip_try(const SrnConvParams& p, hipStream_t stream) {
  if (p.precision != SRN_PREC_BF16X3 || p.w_hi == nullptr || p.w_bs != 0 || p.w_hs != 0 || p.w_nmajor) return 0;
  if (p.n_head != 1 || p.in_stride != 1 || p.pad_reflect || p.geglu) return 0;
  if (!(p.C_in == 32 || p.C_in == 64) || p.C_in0 != p.C_in || !(p.N == 32 || p.N == 64)) return 0;
  if (!(p.pro_act == SRN_ACT_NONE || p.pro_act == SRN_ACT_LEAKY)) return 0;
  const bool force = p.no_halo == 4;  // tests / A-B timing: take every structurally eligible shape
  if (!force && p.T_out < 4 * BM) return 0;  // short sequences: the tiled kernels fill the chip better
  int lo = p.tap_off[0], hi = p.tap_off[0];
  for (int i = 1; i < p.n_taps; ++i) {
    lo = p.tap_off[i] < lo ? p.tap_off[i] : lo;
    hi = p.tap_off[i] > hi ? p.tap_off[i] : hi;
  }
  if (hi - lo > HALO_MAX) return 0;
  const int smem = p.n_taps * p.C_in * p.N * 4 + (p.C_in / 32) * 2 * HR_MAX * 64;
  if (smem > 160 * 1024) return 0;
  // measured on the HiFi-GAN stages (8 x 245760 x 32 and 8 x 122880 x 64): ahead of the tiled kernels with >= 2
  // workgroups per CU (32 channels: k3 0.242 -> 0.212, k7 0.328 -> 0.190, k11 0.355 -> 0.274 ms); with one (64
  // channels: 95-160 KB) the exposed load latency loses (k3 0.201 -> 0.445 ms)
  if (!force && smem > 78 * 1024) return 0;
  if (p.C_in == 32 && p.N == 32) return launch_strip_act<32, 1>(p, lo, hi - lo, stream);
  if (p.C_in == 32 && p.N == 64) return launch_strip_act<32, 2>(p, lo, hi - lo, stream);
  if (p.C_in == 64 && p.N == 32) return launch_strip_act<64, 1>(p, lo, hi - lo, stream);
  return launch_strip_act<64, 2>(p, lo, hi - lo, stream);
}
