// mfma_feed_costs.hip -- stand-alone: what do the FEED instructions of the exact-fp32 loop cost its matrix stream at the
// rate the real loop issues them (per 16 MFMAs of one wave: 4 loads, 4 LDS stores, 8 LDS reads, 2 barriers)?
// Every wave of the chip runs: 16 x v_mfma_f32_32x32x2_f32 with the chosen feed instructions spread between them.
// Reported: TFLOP/s of the MFMA stream and the implied cycles per 16-MFMA step (1024 = free).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE, F_DSW128, F_DSW64, F_DSR, F_BUF, F_BUF_DSW, F_DMA, F_BAR, F_ALL_REG, F_ALL_DMA };

template <int KIND>
__global__ __launch_bounds__(256) void feed_kernel(const float* src, float* out, int iters, float a0, float b0) {
  __shared__ __attribute__((aligned(16))) float lds[8192];  // 32 KB
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a0;
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  const unsigned la = threadIdx.x * 16;  // LDS byte address of this lane's 16-B slot
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 26, 0x00020000);
  const int voff = (blockIdx.x % 64) * 65536 + threadIdx.x * 16;
  int soff = 0;
  f32x4 q[4] = {};
  u32x4 g[4] = {};
#define MFMA4                                                                              \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));      \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));      \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));      \
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (KIND == F_DSW128 || KIND == F_ALL_REG)
        asm volatile("ds_write_b128 %0, %1" : : "v"(la), "v"(q[j]) : "memory");
      if constexpr (KIND == F_DSW64) {
        asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"(*(double*)&q[j]) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:8" : : "v"(la), "v"(*(double*)&q[j]) : "memory");
      }
      if constexpr (KIND == F_DSR || KIND == F_ALL_REG || KIND == F_ALL_DMA) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(q[j]) : "v"(la) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q[(j + 1) & 3]) : "v"(la) : "memory");
      }
      if constexpr (KIND == F_BUF || KIND == F_ALL_REG)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[j]) : "v"(voff), "s"(rs), "s"(soff) : "memory");
      if constexpr (KIND == F_BUF_DSW) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[j]) : "v"(voff), "s"(rs), "s"(soff) : "memory");
        asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %0, %1" : : "v"(la), "v"(g[(j + 1) & 3]) : "memory");
      }
      if constexpr (KIND == F_DMA || KIND == F_ALL_DMA) {
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     : : "s"(wave * 1024 + j * 8192), "v"(voff), "s"(rs), "s"(soff) : "memory");
      }
      MFMA4
      soff += 4096;
      if (soff >= 65536) soff = 0;
    }
    if constexpr (KIND == F_BAR || KIND == F_ALL_REG || KIND == F_ALL_DMA) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int j = 0; j < 4; ++j) s += q[j].x + q[j].w + __builtin_bit_cast(float, g[j].x) + __builtin_bit_cast(float, g[j].w);
  if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int wg_per_cu, const float* src, float* d) {
  const int grid = 256 * wg_per_cu, iters = 2000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  feed_kernel<KIND><<<grid, 256>>>(src, d, 200, 1.f, 2.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  feed_kernel<KIND><<<grid, 256>>>(src, d, iters, 1.f, 2.f);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, s, e);
  const double tf = (double)grid * 4 * iters * 16 * 4096.0 / ms / 1e9;
  std::printf("%d waves/SIMD  %-44s : %6.1f TFLOP/s  %6.0f cycles per 16-MFMA step (at 2.4 GHz, 1024 = free)\n", wg_per_cu, name, tf,
              1024.0 * 157.3 / tf);
}

int main() {
  float *src, *d;
  (void)hipMalloc(&src, 1 << 26);
  (void)hipMemset(src, 0, 1 << 26);
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {2, 4}) {
    run<F_NONE>("nothing", w, src, d);
    run<F_DSW128>("4 ds_write_b128", w, src, d);
    run<F_DSW64>("8 ds_write_b64", w, src, d);
    run<F_DSR>("8 ds_read_b128", w, src, d);
    run<F_BUF>("4 buffer_load_dwordx4 (L2 hits)", w, src, d);
    run<F_BUF_DSW>("4 buffer_load + 4 ds_write_b128", w, src, d);
    run<F_DMA>("4 buffer_load ... lds", w, src, d);
    run<F_BAR>("2 barriers", w, src, d);
    run<F_ALL_REG>("loads + stores + reads + barriers", w, src, d);
    run<F_ALL_DMA>("DMA loads + reads + barriers", w, src, d);
  }
  return 0;
}
