// conv_planes.hip — the split-bf16 contraction as a pure LDS-DMA + MFMA pipeline.
//
// conv_gemm.hip converts every fp32 operand to a (hi, lo) bf16 pair while staging it into LDS, in every workgroup
// and for every tap: that VALU + ds_write work, and the register round trip that forbids deep prefetch, is what
// bounds its split-bf16 mode (~60 % of the wave lifetime is spent waiting, profiles/r1_pmc_*).  Here the operands are
// split ONCE into bf16 planes in a caller-provided HBM workspace (activations: a streaming pass per launch; weights:
// pre-split at load), and the GEMM kernel moves 16-B pieces global -> LDS with `global_load_lds_dwordx4` (no VGPRs,
// no VALU), three pipeline stages deep behind a counted `s_waitcnt vmcnt(N)` and one raw `s_barrier` per K-step:
//
//   step s:  wait own DMA(s) | barrier | issue DMA(s+2) into the stage freed by step s-1 | 24..6 MFMA on stage s
//
// Plane layout (HBM and LDS alike): per row, per 32-channel chunk, 128 B = [32 bf16 hi | 32 bf16 lo], so one DMA
// instruction moves 8 rows x one full 128-B line each (half-line, 64-B-per-row fetches measured ~40 % slower).
// LDS stage = [A rows | B rows] of 128 B whose eight 16-B chunks are XOR-swizzled by (row >> 1) & 7 (conflict-free
// ds_read_b128 for 16 consecutive rows).  The DMA writes LDS linearly (wave base + lane * 16), so the swizzle is
// applied to the per-lane SOURCE address.  Rows that are padding / masked / out of range come from a zero page.
#include "conv_common.h"

namespace {

constexpr int BK = 32;
#ifndef SRN_PLANES_NSTAGE
#define SRN_PLANES_NSTAGE 2
#endif
constexpr int NSTAGE = SRN_PLANES_NSTAGE;  // 3: prefetch distance 2 at 1 block/CU; 2: double buffer at 2 blocks/CU

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// ------------------------------------------------------------------------------------------------
// split passes: fp32 -> (hi, lo) bf16 planes, rows padded to Cp = roundup(C, 32) channels
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in0, int64_t in0_bs, int64_t in0_hs,
                                                         int ld0, int C0, const float* __restrict__ in1,
                                                         int64_t in1_bs, int ld1, int C, int Cp, int T, int n_head,
                                                         int act, float slope, __bf16* __restrict__ pl) {
  const int z = blockIdx.y;
  const int zb = z / n_head, zh = z - zb * n_head;
  const int c8n = Cp / 8;
  const int64_t total = (int64_t)T * c8n;
  const float* a = in0 + (int64_t)zb * in0_bs + (int64_t)zh * in0_hs;
  const float* b = in1 ? in1 + (int64_t)zb * in1_bs : nullptr;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int t = (int)(idx / c8n);
    const int c = (int)(idx - (int64_t)t * c8n) * 8;
    float4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = c + 4 * h;
      v[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cc < C) {
        v[h] = cc < C0 ? *reinterpret_cast<const float4*>(a + (int64_t)t * ld0 + cc)
                       : *reinterpret_cast<const float4*>(b + (int64_t)t * ld1 + (cc - C0));
        if (act == SRN_ACT_LEAKY) v[h] = leaky4(v[h], slope);
        else if (act != SRN_ACT_NONE) {
          v[h].x = srn_act(v[h].x, act, slope);
          v[h].y = srn_act(v[h].y, act, slope);
          v[h].z = srn_act(v[h].z, act, slope);
          v[h].w = srn_act(v[h].w, act, slope);
        }
      }
    }
    bf16x4 h0, l0, h1, l1;
    split4(v[0], h0, l0);
    split4(v[1], h1, l1);
    bf16x8 hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    bf16x8 ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    // element (row, k): hi at row * 2 Cp + (k / 32) * 64 + k % 32, lo 32 elements further
    const int64_t o = ((int64_t)z * T + t) * 2 * Cp + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<bf16x8*>(pl + o) = hh;
    *reinterpret_cast<bf16x8*>(pl + o + 32) = ll;
  }
}

// n-major operand (V of P.V): src [k][n] -> planes [z][n][Kp]  (32 x 32 LDS tile transpose)
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ src, int64_t bs, int64_t hs,
                                                              int ld, int K, int Kp, int N, int n_head,
                                                              __bf16* __restrict__ pl) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z;
  const int zb = z / n_head, zh = z - zb * n_head;
  const float* s = src + (int64_t)zb * bs + (int64_t)zh * hs;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, n = n0 + tx;
    tile[i][tx] = (k < K && n < N) ? s[(int64_t)k * ld + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, k = k0 + tx;
    if (n < N && k < Kp) {
      const float v = tile[tx][i];
      const __bf16 h = (__bf16)v;
      const int64_t o = ((int64_t)z * N + n) * 2 * Kp + (k >> 5) * 64 + (k & 31);
      pl[o] = h;
      pl[o + 32] = (__bf16)(v - (float)h);
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int BM_, int BN_, int WM_, int WN_>
struct PCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int MT = WM / 32, NT = WN / 32;
  static constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  static constexpr int STAGE = (BM + BN) * 128;  // bytes
  static constexpr int SMEM_BYTES = NSTAGE * STAGE;
  static constexpr int A_I = BM / 32;   // DMA instructions per wave per step for A (8 rows x 128 B each)
  static constexpr int B_I = BN / 32;
  static constexpr int PER = A_I + B_I;  // DMA instructions per wave per step
};

struct PlaneArgs {
  const __bf16* a;     // [z][T_in][Cp / 32][hi 32 | lo 32]
  const __bf16* w;     // [z or 1][N][n_taps][Cp / 32][hi 32 | lo 32]
  const __bf16* zero;  // >= 128 B of zeros
  int Cp;              // padded channels per tap
  int64_t w_zs;        // weight plane stride per z (0: shared)
};

template <class C>
__global__ __launch_bounds__(256, NSTAGE == 2 ? 2 : 1) void conv_planes_kernel(const SrnConvParams p, const PlaneArgs q, const int m_tiles,
                                                          const int n_tiles) {
  constexpr int BM = C::BM, BN = C::BN, MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int logical = xcd_logical_block();
  int z, mt_i, nt_i;
  tile_coords(logical, m_tiles, n_tiles, z, mt_i, nt_i);
  const int zb = z / p.n_head;
  const int zh = z - zb * p.n_head;
  const int t0 = mt_i * BM;
  const int n0 = nt_i * BN;

  const int T_in = p.T_in;
  int len_in = T_in;
  if (p.len_in) len_in = min(p.len_in[zb], T_in);
  const int Cp = q.Cp;
  const int n_chunks = Cp / BK;
  const int n_steps = p.n_taps * n_chunks;

  // ---- per-lane DMA coordinates: one instruction = 8 rows x 128 B; lane -> (row lane >> 3, 16-B chunk lane & 7)
  const int lrow = lane >> 3;
  const int pc = lane & 7;
  const unsigned char* a_base = reinterpret_cast<const unsigned char*>(q.a) + (int64_t)z * T_in * Cp * 4;
  int a_tb[C::A_I];  // input row before the tap offset (or very negative if the output row is padding)
  int a_lc[C::A_I];  // logical chunk this lane fetches (source-side swizzle)
#pragma unroll
  for (int i = 0; i < C::A_I; ++i) {
    const int r = 8 * (wave + 4 * i) + lrow;
    const int t = t0 + r;
    a_tb[i] = t < p.T_out ? t * p.in_stride : -(1 << 29);
    a_lc[i] = pc ^ ((r >> 1) & 7);
  }
  const unsigned char* b_ptr[C::B_I];  // row pointer incl. logical chunk, without the (tap, chunk) offset
  bool b_ok[C::B_I];
#pragma unroll
  for (int i = 0; i < C::B_I; ++i) {
    const int r = 8 * (wave + 4 * i) + lrow;
    const int n = n0 + r;
    b_ok[i] = n < p.N;
    const int lc = pc ^ ((r >> 1) & 7);
    b_ptr[i] = reinterpret_cast<const unsigned char*>(q.w) + (int64_t)z * q.w_zs * 2 +
               ((int64_t)min(n, p.N - 1) * p.n_taps * Cp * 2 + lc * 8) * 2;
  }
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(q.zero) + pc * 16;

  auto issue = [&](const int step, const int stage) {
    const int tap = step / n_chunks;
    const int chunk = step - tap * n_chunks;
    const int toff = p.tap_off[tap];
    unsigned char* sbase = smem_p + stage * C::STAGE;
#pragma unroll
    for (int i = 0; i < C::A_I; ++i) {
      int ti = a_tb[i] + toff;
      if (p.pad_reflect) {
        if (ti < 0 && ti > -(1 << 28)) ti = -ti;
        if (ti >= T_in) ti = 2 * (T_in - 1) - ti;
      }
      const bool ok = ti >= 0 && ti < len_in;
      const unsigned char* src = ok ? a_base + ((int64_t)ti * Cp * 2 + chunk * 64 + a_lc[i] * 8) * 2 : zero;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sbase + (wave + 4 * i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < C::B_I; ++i) {
      const unsigned char* src = b_ok[i] ? b_ptr[i] + ((int64_t)tap * Cp * 2 + chunk * 64) * 2 : zero;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(sbase + BM * 128 + (wave + 4 * i) * 1024), 16,
                                       0, 0);
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
  const int sw = (li >> 1) & 7;  // row swizzle key (tile row offsets are multiples of 32)

  auto compute = [&](const int stage) {
    const unsigned char* a_s = smem_p + stage * C::STAGE;
    const unsigned char* b_s = a_s + BM * 128;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int c_hi = (((kk * 2 + lh) ^ sw) & 7) << 4;      // 16-B chunk of the hi half, swizzled
      const int c_lo = (((kk * 2 + lh + 4) ^ sw) & 7) << 4;  // same k range in the lo half
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int o = (wm0 + m * 32 + li) * 128;
        ah[m] = *reinterpret_cast<const bf16x8*>(a_s + o + c_hi);
        al[m] = *reinterpret_cast<const bf16x8*>(a_s + o + c_lo);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int o = (wn0 + n * 32 + li) * 128;
        bh[n] = *reinterpret_cast<const bf16x8*>(b_s + o + c_hi);
        bl[n] = *reinterpret_cast<const bf16x8*>(b_s + o + c_lo);
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

  // ---- pipeline
  if constexpr (NSTAGE == 2) {
    // double buffer: DMA(step+1) flies during compute(step); the co-resident block fills the wait
    issue(0, 0);
    for (int step = 0; step < n_steps; ++step) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everyone's DMA(step) landed; everyone is done reading the other stage
      if (step + 1 < n_steps) issue(step + 1, (step + 1) & 1);
      compute(step & 1);
    }
  } else {
    // prefetch distance 2, three LDS stages
    issue(0, 0);
    if (n_steps > 1) issue(1, 1);
    int stage = 0;
    for (int step = 0; step < n_steps; ++step) {
      // own DMA(step) has landed once at most the PER instructions of DMA(step+1) are still outstanding
      if (step + 1 < n_steps) {
        if constexpr (C::PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (C::PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (C::PER == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (C::PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // everyone's DMA(step) landed; everyone is done reading the stage of step-1
      if (step + 2 < n_steps) {
        int s2 = stage + 2;
        if (s2 >= NSTAGE) s2 -= NSTAGE;
        issue(step + 2, s2);
      }
      compute(stage);
      if (++stage == NSTAGE) stage = 0;
    }
  }

  conv_epilogue<MT, NT>(p, acc, zb, zh, t0, n0, wm0, wn0, lane);
}

template <class C>
int launch_planes(const SrnConvParams& p, const PlaneArgs& q, hipStream_t stream) {
  constexpr int SMEM = C::SMEM_BYTES;
  static SrnSmemAttr smem_attr;
  if (const int e = smem_attr.ensure(reinterpret_cast<const void*>(&conv_planes_kernel<C>), SMEM)) return e;
  const int m_tiles = (p.T_out + C::BM - 1) / C::BM;
  const int n_tiles = (p.N + C::BN - 1) / C::BN;
  const int64_t blocks = (int64_t)p.n_batch * p.n_head * m_tiles * n_tiles;
  SRN_CHECK_ARG(blocks > 0 && blocks < (1ll << 31), "conv_planes: bad grid %lld", (long long)blocks);
  hipLaunchKernelGGL((conv_planes_kernel<C>), dim3((unsigned)blocks), dim3(256), SMEM, stream, p, q, m_tiles, n_tiles);
  SRN_CHECK_LAUNCH();
  return 1;
}

inline int64_t rup64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace

// Returns 1 if the launch was handled, 0 if not eligible (caller falls back to conv_gemm), < 0 on error.
int srn_conv_planes_try(const SrnConvParams& p, int tile, hipStream_t stream) {
  if (p.precision != SRN_PREC_BF16X3 || p.ws == nullptr || p.geglu) return 0;
  const int Z = p.n_batch * p.n_head;
  const int Cp = (int)rup64(p.C_in, 32);
  const int64_t a_elems = (int64_t)Z * p.T_in * Cp * 2;          // hi + lo
  const int64_t w_elems_z = (int64_t)p.N * p.n_taps * Cp * 2;    // per z
  const bool w_ready = p.w_hi != nullptr;
  const bool w_per_z = p.w_bs != 0 || p.w_hs != 0;
  const int64_t w_total = w_ready ? 0 : w_elems_z * (w_per_z ? Z : 1);
  const int64_t need = 256 + a_elems * 2 + w_total * 2 + 64;
  if (p.ws_bytes < need) return 0;
  if (!w_ready && !p.w_nmajor && p.n_taps != 1) return 0;  // multi-tap weights must be pre-split at load time
  unsigned char* ws = reinterpret_cast<unsigned char*>(p.ws);
  SRN_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "conv_planes: workspace must be 256-byte aligned");
  // [zero page 256 B][A planes][W planes]
  SRN_CHECK_HIP(hipMemsetAsync(ws, 0, 256, stream));
  __bf16* a_pl = reinterpret_cast<__bf16*>(ws + 256);
  __bf16* w_pl = a_pl + a_elems;
  {
    const int64_t total = (int64_t)p.T_in * (Cp / 8);
    int64_t bx = (total + 255) / 256;
    if (bx > 4096) bx = 4096;
    const int C0 = (p.C_in0 > 0 && p.C_in0 < p.C_in) ? p.C_in0 : p.C_in;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)bx, Z), dim3(256), 0, stream, p.in0, p.in0_bs, p.in0_hs,
                       p.ld_in0, C0, p.in1, p.in1_bs, p.ld_in1, p.C_in, Cp, p.T_in, p.n_head, p.pro_act, p.pro_slope,
                       a_pl);
    SRN_CHECK_LAUNCH();
  }
  PlaneArgs q;
  q.a = a_pl;
  q.zero = reinterpret_cast<const __bf16*>(ws);
  q.Cp = Cp;
  if (w_ready) {
    q.w = reinterpret_cast<const __bf16*>(p.w_hi);
    q.w_zs = 0;
  } else {
    const int Zw = w_per_z ? Z : 1;
    const int nh = w_per_z ? p.n_head : 1;
    const int Cw = p.C_w > 0 ? p.C_w : p.C_in;
    if (p.w_nmajor) {
      dim3 grid((Cp + 31) / 32, (p.N + 31) / 32, Zw);
      hipLaunchKernelGGL(split_transpose_kernel, grid, dim3(256), 0, stream, p.w, p.w_bs, p.w_hs, p.ldw, Cw, Cp, p.N,
                         nh, w_pl);
    } else {
      const int64_t total = (int64_t)p.N * (Cp / 8);
      int64_t bx = (total + 255) / 256;
      if (bx > 4096) bx = 4096;
      hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)bx, Zw), dim3(256), 0, stream, p.w, p.w_bs, p.w_hs, p.ldw,
                         Cw, (const float*)nullptr, (int64_t)0, 0, Cw, Cp, p.N, nh, (int)SRN_ACT_NONE, 0.f, w_pl);
    }
    SRN_CHECK_LAUNCH();
    q.w = w_pl;
    q.w_zs = w_per_z ? w_elems_z : 0;
  }
  switch (tile) {
    case 1: return launch_planes<PCfg<128, 128, 64, 64>>(p, q, stream);
    case 2: return launch_planes<PCfg<128, 64, 32, 64>>(p, q, stream);
    case 3: return launch_planes<PCfg<64, 128, 32, 64>>(p, q, stream);
    case 4: return launch_planes<PCfg<64, 64, 32, 32>>(p, q, stream);
    case 5: return launch_planes<PCfg<128, 32, 32, 32>>(p, q, stream);
    default: return 0;
  }
}
