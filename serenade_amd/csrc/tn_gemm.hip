// tn_gemm.hip -- contractions over TIME for the training step: weight gradients of every conv / linear
// (dW_j = sum_{b,t} dY[b,t,:]^T X[b, t*stride + o_j, :]) and the two attention gradients that contract over query
// rows (dV = P^T dO, dK = dS^T Q), exact fp32 on v_mfma_f32_32x32x2_f32.  Reference: what autograd does for
// serenade/models/matcha_components/decoder.py's Conv1d / Linear layers and diffusers' attention under
// trainers/ssc.py:57-96 (loss.backward()).
//
// Both operands are TIME-MAJOR in HBM (row = a time step, channels contiguous), which is exactly the MFMA's fragment
// order once a k-slab sits in LDS as [k][m]: lane (m = lane % 32, k = lane / 32) reads consecutive floats of one row
// per 32-lane group (conflict-free ds_read_b32, no transpose anywhere).  So a block streams 16-row slabs of A and of
// the tap-shifted B (rows outside the item read as zero: "same" padding without guard rows or padded copies) through
// a double-buffered LDS stage with coalesced 16-B global loads, and multiplies them as they are.
//
// The output (M x N per tap) is small and the contraction long (batch x time), so the time axis is sliced over
// `ksplit` workgroup sets writing raw partial tiles to a workspace; srn_tn_gemm's reduce kernel adds the slices IN
// SLICE ORDER (bit-reproducible, no atomics) and scales.  Attention gradients (one problem per batch x head, K = L)
// fill the chip without slicing and store directly.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"
#include "serenade_hip.h"

namespace {

constexpr int MAX_SPLIT = 32;
constexpr int BK = 16;  // rows per LDS slab (32 measured no faster: 77.5 vs 78.2 TFLOP/s overall)

// TB = tile edge (128 or 64): 4 waves, each a (TB/2) x (TB/2) block of 32 x 32 MFMA tiles.
template <int TB>
__global__ __launch_bounds__(256, 2) void tn_gemm_kernel(const SrnTnGemmParams p, const int m_tiles, const int n_tiles,
                                                          const int ksplit, const int k_per) {
  constexpr int BM = TB, BN = TB;
  // floats per k-row of a slab: 16-B aligned rows, banks rotate by 4 per row; unpadded where the padded pair of
  // double-buffered slabs would pass the 64 KB static limit (a half-wave reads one row: still conflict-free)
  constexpr int PITCH = (2 * 2 * BK * (TB + 4) * 4 > 65536) ? TB : TB + 4;
  constexpr int SLAB = BK * PITCH;
  constexpr int WT = TB / 64;         // 32 x 32 tiles per wave per direction
  constexpr int F4 = TB / 4;          // float4 pieces per slab row
  constexpr int ROWS_PER_PASS = 256 / F4;
  constexpr int NLD = BK / ROWS_PER_PASS;  // float4 loads per thread per operand per step (2 at 128, 1 at 64)
  __shared__ __attribute__((aligned(16))) float lds[2][2][SLAB];  // [stage][A | B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid: x = tile (m fastest) of one (problem, tap), y = tap, z = problem * ksplit + slice
  const int mt = blockIdx.x % m_tiles, nt = blockIdx.x / m_tiles;
  const int tap = blockIdx.y;
  const int z = blockIdx.z / ksplit, slice = blockIdx.z - z * ksplit;
  const int zb = z / p.n_head, zh = z - zb * p.n_head;
  const int m0 = mt * BM, n0 = nt * BN;
  const int shift = p.shift[tap];
  const float* A = p.a + (int64_t)zb * p.a_bs + (int64_t)zh * p.a_hs;
  const float* Bm = p.b + (int64_t)zb * p.b_bs + (int64_t)zh * p.b_hs;
  const int K = p.n_items * p.T_a;  // contraction index r = item * T_a + t
  const int k_begin = slice * k_per, k_end = min(K, k_begin + k_per);
  const int n_inner = p.n_inner > 1 ? p.n_inner : 1;

  // global -> register -> LDS: per step each operand is 16 rows x TB floats
  const int lrow = tid / F4;
  const int lcol = (tid % F4) * 4;  // float offset in the slab row
  float4 ra[NLD], rb[NLD];
  // per load slot: the contraction row r = item * T_a + t it fetches next, kept as (t, row pointers) and advanced by BK
  // rows per step -- the divisions that split r happen once here and again only when a slot crosses into the next item
  int lr[NLD], lt[NLD], litem[NLD], lend[NLD];  // lend: rows of b at or past it read as zero (T_b, or the item's len_b)
  const float *pa[NLD], *pb[NLD];
  auto seat = [&](const int i) {  // pointers of slot i at (litem, lt)
    const int i1 = litem[i] / n_inner, i2 = litem[i] - i1 * n_inner;  // items may be a 2-level grid (conv2d: batch x row)
    pa[i] = A + (int64_t)i1 * p.a_is + (int64_t)i2 * p.a_is2 + (int64_t)lt[i] * p.lda + m0 + lcol;
    pb[i] = Bm + (int64_t)i1 * p.b_is + (int64_t)i2 * p.b_is2 + ((int64_t)lt[i] * p.stride + shift) * p.ldb + n0 + lcol;
    lend[i] = p.len_b != nullptr && litem[i] < p.n_items ? min(p.T_b, p.len_b[zb * p.n_items + litem[i]]) : p.T_b;
  };
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    lr[i] = k_begin + lrow + ROWS_PER_PASS * i;
    litem[i] = lr[i] / p.T_a;
    lt[i] = lr[i] - litem[i] * p.T_a;
    seat(i);
  }
  const bool a_in = m0 + lcol < p.M, b_in = n0 + lcol < p.N;
  const int64_t a_step = (int64_t)BK * p.lda, b_step = (int64_t)BK * p.stride * p.ldb;
  auto load = [&]() {  // fetch the slots' rows, then advance them one step
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (lr[i] < k_end) {
        if (a_in) va = *reinterpret_cast<const float4*>(pa[i]);
        const int tb = lt[i] * p.stride + shift;
        if (tb >= 0 && tb < lend[i] && b_in) vb = *reinterpret_cast<const float4*>(pb[i]);
      }
      ra[i] = va;
      rb[i] = vb;
      lr[i] += BK;
      lt[i] += BK;
      pa[i] += a_step;
      pb[i] += b_step;
      if (lt[i] >= p.T_a) {
        do {
          lt[i] -= p.T_a;
          ++litem[i];
        } while (lt[i] >= p.T_a);
        seat(i);
      }
    }
  };
  auto store = [&](const int st) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<float4*>(&lds[st][0][(lrow + ROWS_PER_PASS * i) * PITCH + lcol]) = ra[i];
      *reinterpret_cast<float4*>(&lds[st][1][(lrow + ROWS_PER_PASS * i) * PITCH + lcol]) = rb[i];
    }
  };

  const int wm0 = (wave >> 1) * (TB / 2), wn0 = (wave & 1) * (TB / 2);
  const int fm = lane & 31, fk = lane >> 5;
  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // column sums of a (a conv's bias gradient) ride along in the workgroups of the first column of tiles: thread m adds
  // the slab's BK values of column m after the slab's barrier -- 16 conflict-free LDS reads beside 32 MFMAs
  const bool do_cs = p.colsum != nullptr && nt == 0 && tap == 0;
  float csum = 0.f;
  const int n_steps = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;
  if (n_steps > 0) {
    load();
    store(0);
  }
  __syncthreads();
  for (int s = 0; s < n_steps; ++s) {
    const int cur = s & 1;
    if (s + 1 < n_steps) load();
    const float* la = &lds[cur][0][0];
    const float* lb = &lds[cur][1][0];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int row = (kk + fk) * PITCH;
      float av[WT], bv[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        av[i] = la[row + wm0 + 32 * i + fm];
        bv[i] = lb[row + wn0 + 32 * i + fm];
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (do_cs && tid < TB) {
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) csum += la[kk * PITCH + tid];
    }
    if (s + 1 < n_steps) store(cur ^ 1);
    __syncthreads();
  }
  if (do_cs && tid < TB && m0 + tid < p.M) {
    if (ksplit > 1)
      p.ws[(int64_t)ksplit * p.n_batch * p.n_head * p.M * p.n_shifts * p.N + (int64_t)slice * p.M + m0 + tid] = csum;
    else
      p.colsum[m0 + tid] = csum * p.alpha;
  }

  // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float* out;
  int64_t ld;
  float alpha = 1.f;
  if (ksplit > 1) {
    ld = (int64_t)p.n_shifts * p.N;
    out = p.ws + (((int64_t)slice * p.n_batch * p.n_head + z) * p.M) * ld + (int64_t)tap * p.N;
  } else {
    ld = p.ldc;
    out = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs + (int64_t)tap * p.N;
    alpha = p.alpha;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int col = n0 + wn0 + 32 * j + (lane & 31);
      if (col >= p.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < p.M) out[(int64_t)row * ld + col] = acc[i][j][e] * alpha;
      }
    }
}

// The same kernel with its operand walk on the scalar unit (round 4): v_mfma_f32_32x32x2_f32 shares the SIMD's fp32 lanes
// with ordinary vector instructions (conv_f32.hip), and the general kernel above spends ~80 of them per 32-MFMA step on
// 64-bit row pointers, bounds compares and selects.  When every 16-row slab lies inside ONE item (T_a % 16 == 0, one-level
// items) the item, the row inside it and both tensors' descriptors are wave-uniform: A is fetched with a constant per-lane
// offset and the row in the scalar offset, B with one add per load (its row must sit in the per-lane offset: rows outside
// [0, len_b) then fall outside the item's descriptor and read as zeros -- "same" padding and the x * mask without a compare).
// LDS image, fragment reads, MFMA order, column sums and epilogue are the general kernel's: results agree bit for bit.
template <int TB>
__global__ __launch_bounds__(256, 2) void tn_lean_kernel(const SrnTnGemmParams p, const int m_tiles, const int n_tiles,
                                                         const int ksplit, const int k_per) {
  constexpr int BM = TB, BN = TB;
  constexpr int PITCH = (2 * 2 * BK * (TB + 4) * 4 > 65536) ? TB : TB + 4;
  constexpr int SLAB = BK * PITCH;
  constexpr int WT = TB / 64;
  constexpr int F4 = TB / 4;
  constexpr int ROWS_PER_PASS = 256 / F4;
  constexpr int NLD = BK / ROWS_PER_PASS;
  typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float lds[2][2][SLAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x / m_tiles, mt = blockIdx.x - nt * m_tiles;  // (scalar divisions: once per workgroup)
  const int tap = blockIdx.y;
  const int z = blockIdx.z / ksplit, slice = blockIdx.z - z * ksplit;
  const int zb = z / p.n_head, zh = z - zb * p.n_head;
  const int m0 = mt * BM, n0 = nt * BN;
  const int shift = p.shift[tap];
  const float* A = p.a + (int64_t)zb * p.a_bs + (int64_t)zh * p.a_hs;
  const float* Bm = p.b + (int64_t)zb * p.b_bs + (int64_t)zh * p.b_hs;
  const int K = p.n_items * p.T_a;
  const int k_begin = slice * k_per, k_end = min(K, k_begin + k_per);
  const int n_steps = k_end > k_begin ? (k_end - k_begin) / BK : 0;

  const int lrow = tid / F4;
  const int lcol = (tid % F4) * 4;
  const unsigned OOBT = 0x80000000u;
  // the next slab to fetch: item, first row inside it (wave-uniform); descriptors of that item
  int item = k_begin / p.T_a;
  int t0 = k_begin - item * p.T_a;
  const int a_bytes = ((p.T_a - 1) * p.lda + (p.M + 3) / 4 * 4) * 4;
  unsigned voff_a[NLD], voff_b[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i)
    voff_a[i] = m0 + lcol < p.M ? (unsigned)(((lrow + ROWS_PER_PASS * i) * p.lda + m0 + lcol) * 4) : OOBT;
  const int b_row_step = BK * p.stride * p.ldb * 4;
  auto seat_b = [&]() {  // per-lane offsets of b's rows for the slab at (item, t0)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int tb = (t0 + lrow + ROWS_PER_PASS * i) * p.stride + shift;
      voff_b[i] = n0 + lcol < p.N ? (unsigned)((tb * p.ldb + n0 + lcol) * 4) : OOBT;  // tb < 0 wraps past 2^31: zeros
    }
  };
  seat_b();
  u32x4t ra[NLD], rb[NLD];
  auto load = [&]() {  // fetch the slab at (item, t0), then advance one slab
    const int it = min(item, p.n_items - 1);
    const int lend = p.len_b != nullptr ? min(p.T_b, p.len_b[zb * p.n_items + it]) : p.T_b;
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (int64_t)it * p.a_is), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bm + (int64_t)it * p.b_is), 0, lend * p.ldb * 4, 0x00020000);
    const int soff_a = t0 * p.lda * 4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff_a[i], soff_a, 0);
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, voff_b[i], 0, 0);
    }
    t0 += BK;
    if (t0 >= p.T_a) {
      t0 = 0;
      ++item;
      seat_b();
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i) voff_b[i] += b_row_step;
    }
  };
  auto store = [&](const int st) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      *reinterpret_cast<u32x4t*>(&lds[st][0][(lrow + ROWS_PER_PASS * i) * PITCH + lcol]) = ra[i];
      *reinterpret_cast<u32x4t*>(&lds[st][1][(lrow + ROWS_PER_PASS * i) * PITCH + lcol]) = rb[i];
    }
  };

  const int wm0 = (wave >> 1) * (TB / 2), wn0 = (wave & 1) * (TB / 2);
  const int fm = lane & 31, fk = lane >> 5;
  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool do_cs = p.colsum != nullptr && nt == 0 && tap == 0;
  float csum = 0.f;
  if (n_steps > 0) {
    load();
    store(0);
  }
  __syncthreads();
  for (int s = 0; s < n_steps; ++s) {
    const int cur = s & 1;
    if (s + 1 < n_steps) load();
    const float* la = &lds[cur][0][0];
    const float* lb = &lds[cur][1][0];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int row = (kk + fk) * PITCH;
      float av[WT], bv[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        av[i] = la[row + wm0 + 32 * i + fm];
        bv[i] = lb[row + wn0 + 32 * i + fm];
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (do_cs && tid < TB) {
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) csum += la[kk * PITCH + tid];
    }
    if (s + 1 < n_steps) store(cur ^ 1);
    __syncthreads();
  }
  if (do_cs && tid < TB && m0 + tid < p.M) {
    if (ksplit > 1)
      p.ws[(int64_t)ksplit * p.n_batch * p.n_head * p.M * p.n_shifts * p.N + (int64_t)slice * p.M + m0 + tid] = csum;
    else
      p.colsum[m0 + tid] = csum * p.alpha;
  }

  float* out;
  int64_t ld;
  float alpha = 1.f;
  if (ksplit > 1) {
    ld = (int64_t)p.n_shifts * p.N;
    out = p.ws + (((int64_t)slice * p.n_batch * p.n_head + z) * p.M) * ld + (int64_t)tap * p.N;
  } else {
    ld = p.ldc;
    out = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs + (int64_t)tap * p.N;
    alpha = p.alpha;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int col = n0 + wn0 + 32 * j + (lane & 31);
      if (col >= p.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < p.M) out[(int64_t)row * ld + col] = acc[i][j][e] * alpha;
      }
    }
}

// out[z][m][:] = alpha * sum_s ws[s][z][m][:], slices in order; one float4 per thread
__global__ __launch_bounds__(256) void tn_reduce_kernel(const SrnTnGemmParams p, const int ksplit) {
  const int64_t cols = (int64_t)p.n_shifts * p.N;  // % 4 == 0
  const int64_t per_z = (int64_t)p.M * cols;
  const int64_t Z = (int64_t)p.n_batch * p.n_head;
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= Z * per_z) {  // the blocks past the tiles: column sums (M % 4 == 0, Z == 1), slices in order too
    const int64_t m = i4 - Z * per_z;
    if (p.colsum == nullptr || m >= p.M) return;
    const float* w = p.ws + (int64_t)ksplit * Z * per_z + m;
    float4 v = *reinterpret_cast<const float4*>(w);
    for (int s = 1; s < ksplit; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(w + (int64_t)s * p.M);
      v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
    }
    *reinterpret_cast<float4*>(p.colsum + m) = make_float4(v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha);
    return;
  }
  const int64_t z = i4 / per_z, rem = i4 - z * per_z;
  const int64_t m = rem / cols, c = rem - m * cols;
  const float* w = p.ws + i4;
  float4 v = *reinterpret_cast<const float4*>(w);
  for (int s = 1; s < ksplit; ++s) {
    const float4 q = *reinterpret_cast<const float4*>(w + (int64_t)s * Z * per_z);
    v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
  }
  const int zb = (int)(z / p.n_head), zh = (int)(z - (int64_t)zb * p.n_head);
  float* o = p.out + (int64_t)zb * p.out_bs + (int64_t)zh * p.out_hs + m * p.ldc + c;
  *reinterpret_cast<float4*>(o) = make_float4(v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha);
}

int plan_split(const SrnTnGemmParams& p, int TB, int& k_per) {
  const int64_t K = (int64_t)p.n_items * p.T_a;
  const int64_t tiles = (int64_t)((p.M + TB - 1) / TB) * ((p.N + TB - 1) / TB) * p.n_shifts * p.n_batch * p.n_head;
  int ks = 1;
  // ~3 workgroups per CU wanted, never fewer than 8 slabs per slice.  Swept on the training step's shapes
  // (tools/tnbench.py, B = 4 x L = 1024): targets 256 / 384 / 512 / 768 / 1024 give 57 / 67 / 75 / 79 / 77 TFLOP/s overall
  static const int target = getenv("SRN_TN_TARGET") ? atoi(getenv("SRN_TN_TARGET")) : 768;
  if (tiles < target / 2) {
    ks = (int)((target + tiles - 1) / tiles);
    const int64_t cap = K / (8 * BK);
    if (ks > cap) ks = (int)cap;
    if (ks > MAX_SPLIT) ks = MAX_SPLIT;
    if (ks < 1) ks = 1;
  }
  k_per = (int)(((K + ks - 1) / ks + BK - 1) / BK * BK);
  ks = (int)((K + k_per - 1) / k_per);  // no empty slice
  return ks < 1 ? 1 : ks;
}

// tile edge.  Measured on the training step's shapes (tools/tnbench.py, B = 4 x L = 1024; SRN_TN_TILE forces one):
// 64 wins only where 128-tiles leave the chip nearly empty even after slicing -- the two narrow weight gradients
// (512 x 256: 27 vs 32 us, 80 x 512: 22 vs 31 us) and short contractions that cannot be sliced further
// (attention dV at L = 512: 62 vs 71 us); everywhere else 128 is 8-40 % faster (twice the MFMA work per LDS byte)
// partial tiles of every slice, then (with colsum) the slices' partial column sums
int64_t ws_floats(const SrnTnGemmParams& p, int ks) {
  return (int64_t)ks * p.n_batch * p.n_head * p.M * p.n_shifts * p.N + (p.colsum != nullptr ? (int64_t)ks * p.M : 0);
}

int tile_edge(const SrnTnGemmParams& p) {
  static const int forced = getenv("SRN_TN_TILE") ? atoi(getenv("SRN_TN_TILE")) : 0;
  if (forced == 64 || forced == 128) return forced;
  const int64_t tiles128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.n_shifts * p.n_batch * p.n_head;
  const int64_t K = (int64_t)p.n_items * p.T_a;
  if (tiles128 <= 8) return 64;
  if (K <= 512 && tiles128 < 384) return 64;
  return 128;
}

}  // namespace

extern "C" int64_t srn_tn_gemm_workspace_bytes(const SrnTnGemmParams* p) {
  if (p == nullptr || p->M <= 0 || p->N <= 0 || p->n_items <= 0 || p->T_a <= 0 || p->n_shifts <= 0) return 0;
  int k_per = 0;
  const int TB = tile_edge(*p);
  const int ks = plan_split(*p, TB, k_per);
  if (ks <= 1) return 0;
  return ws_floats(*p, ks) * (int64_t)sizeof(float);
}

extern "C" int srn_tn_gemm(const SrnTnGemmParams* pp, void* stream_) {
  SRN_CHECK_ARG(pp != nullptr, "tn_gemm: null params");
  const SrnTnGemmParams& p = *pp;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  SRN_CHECK_ARG(p.a && p.b && p.out, "tn_gemm: null a / b / out");
  SRN_CHECK_ARG(p.n_batch > 0 && p.n_head > 0 && p.n_items > 0 && p.T_a > 0 && p.T_b > 0 && p.M > 0 && p.N > 0,
                "tn_gemm: bad sizes");
  SRN_CHECK_ARG(p.n_shifts >= 1 && p.n_shifts <= SRN_MAX_TAPS && p.stride >= 1, "tn_gemm: n_shifts %d / stride %d",
                p.n_shifts, p.stride);
  // M may be ragged (attention: M = L): a's rows are read in 16-byte pieces up to roundup(M, 4) <= lda, and whatever
  // sits in those pad columns only reaches output rows >= M, which are not stored
  SRN_CHECK_ARG(p.N % 4 == 0 && p.lda % 4 == 0 && p.lda >= (p.M + 3) / 4 * 4 && p.ldb % 4 == 0 && p.ldc % 4 == 0 &&
                    p.a_bs % 4 == 0 && p.a_hs % 4 == 0 && p.a_is % 4 == 0 && p.b_bs % 4 == 0 && p.b_hs % 4 == 0 &&
                    p.b_is % 4 == 0 && p.a_is2 % 4 == 0 && p.b_is2 % 4 == 0 && p.out_bs % 4 == 0 && p.out_hs % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(p.a) | reinterpret_cast<uintptr_t>(p.b) |
                      reinterpret_cast<uintptr_t>(p.out)) & 15) == 0,
                "tn_gemm: N, leading dimensions and strides must be multiples of 4 floats (lda >= roundup(M, 4)), pointers "
                "16-byte aligned");
  SRN_CHECK_ARG(p.ldc >= p.n_shifts * p.N, "tn_gemm: ldc %d < n_shifts * N", p.ldc);
  SRN_CHECK_ARG((int64_t)p.n_items * p.T_a < (1ll << 31), "tn_gemm: contraction too long");
  SRN_CHECK_ARG(p.colsum == nullptr || (p.n_batch == 1 && p.n_head == 1 && p.M % 4 == 0 &&
                                        (reinterpret_cast<uintptr_t>(p.colsum) & 15) == 0),
                "tn_gemm: colsum needs one problem (n_batch = n_head = 1), M %% 4 == 0 and a 16-byte aligned pointer");
  SRN_CHECK_ARG(p.len_b == nullptr || p.n_inner <= 1, "tn_gemm: len_b is per item of a one-level item grid");
  int k_per = 0;
  const int TB = tile_edge(p);
  int ks = plan_split(p, TB, k_per);
  if (ks > 1) {
    const int64_t need = ws_floats(p, ks) * (int64_t)sizeof(float);
    if (p.ws == nullptr || p.ws_bytes < need || (reinterpret_cast<uintptr_t>(p.ws) & 15) != 0) {
      ks = 1;  // no (or too small a) workspace: correct, just fewer workgroups
      k_per = (int)(((int64_t)p.n_items * p.T_a + BK - 1) / BK * BK);
    }
  }
  const int m_tiles = (p.M + TB - 1) / TB, n_tiles = (p.N + TB - 1) / TB;
  const int64_t gz = (int64_t)p.n_batch * p.n_head * ks;
  SRN_CHECK_ARG(gz <= 65535 && (int64_t)m_tiles * n_tiles < (1ll << 31), "tn_gemm: grid too large");
  const dim3 grid(m_tiles * n_tiles, p.n_shifts, (unsigned)gz);
  // the scalar-walk form: every slab inside one item, 32-bit byte offsets inside an item (SRN_TN_GENERAL=1: A-B timing
  // and the bit-identity test keep the general kernel)
  const char* const env_general = getenv("SRN_TN_GENERAL");
  const bool general_only = env_general != nullptr && env_general[0] == '1';
  const bool lean = !general_only && p.n_inner <= 1 && p.T_a % BK == 0 && k_per % BK == 0 &&
                    (int64_t)p.T_a * p.lda * 4 < 0x7fffffffll &&
                    ((int64_t)p.T_b + (int64_t)p.T_a * p.stride + 64) * p.ldb * 4 < 0x7fffffffll;
  auto kern = lean ? (TB == 64 ? tn_lean_kernel<64> : tn_lean_kernel<128>)
                   : (TB == 64 ? tn_gemm_kernel<64> : tn_gemm_kernel<128>);
  hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, p, m_tiles, n_tiles, ks, k_per);
  if (ks > 1) {
    const int64_t n4 = (int64_t)p.n_batch * p.n_head * p.M * p.n_shifts * p.N / 4 + (p.colsum != nullptr ? p.M / 4 : 0);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, p, ks);
  }
  SRN_CHECK_LAUNCH();
  return 0;
}
