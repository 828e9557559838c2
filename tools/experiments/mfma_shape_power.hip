// mfma_shape_power.hip -- stand-alone: does the exact-fp32 loop clock higher at the board's power cap on
// v_mfma_f32_16x16x4_f32 (4 accumulator registers read + written per 2048 FLOPs) than on v_mfma_f32_32x32x2_f32 (16 per
// 4096)?  Every wave runs the real loop's feed per 32-deep step (4 buffer loads of L2-resident lines, 4 ds_write_b128,
// 8 ds_read_b128, 2 barriers) and the step's MFMAs in one of the two shapes; launches last ~100 ms so the power
// management settles.  Reported: TFLOP/s of the MFMA stream (157.3 = 2.4 GHz, nothing in the way).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool FEED>
__global__ __launch_bounds__(256) void k(const float* src, float* out, int iters, float a0, float b0) {
  __shared__ __attribute__((aligned(16))) float lds[4608];  // 18 KB: the 64 x 64 tile's stage
  for (int i = threadIdx.x; i < 4608; i += 256) lds[i] = a0;
  __syncthreads();
  f32x16 acc;
  f32x4 c0, c1, c2, c3;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int e = 0; e < 4; ++e) c0[e] = c1[e] = c2[e] = c3[e] = 0.f;
  const unsigned la = (threadIdx.x % 256) * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 26, 0x00020000);
  const int voff = (blockIdx.x % 64) * 65536 + threadIdx.x * 16;
  int soff = 0;
  f32x4 q[4] = {};
  u32x4 g[4] = {};
  for (int e = 0; e < 4; ++e) q[e].x = a0 + threadIdx.x * 1e-3f, q[e].y = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (FEED) {
        asm volatile("ds_write_b128 %0, %1" : : "v"(la), "v"(g[j]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(q[j]) : "v"(la) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(q[(j + 1) & 3]) : "v"(la) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[j]) : "v"(voff), "s"(rs), "s"(soff) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (SHAPE == 32) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(q[j].x), "v"(q[j].y));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(q[j].y), "v"(q[j].z));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(q[j].z), "v"(q[j].w));
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(q[j].w), "v"(q[j].x));
      } else {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(q[j].x), "v"(q[j].y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(q[j].y), "v"(q[j].z));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c2) : "v"(q[j].z), "v"(q[j].w));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(q[j].w), "v"(q[j].x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(q[j].y), "v"(q[j].x));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(q[j].z), "v"(q[j].y));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c2) : "v"(q[j].w), "v"(q[j].z));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(q[j].x), "v"(q[j].w));
      }
      soff += 4096;
      if (soff >= 65536) soff = 0;
    }
    if constexpr (FEED) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  for (int j = 0; j < 4; ++j) s += q[j].x + q[j].w + __builtin_bit_cast(float, g[j].x) + __builtin_bit_cast(float, g[j].w);
  if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, bool FEED>
void run(const char* name, int wg_per_cu, const float* src, float* d) {
  const int grid = 256 * wg_per_cu, iters = 60000;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(s);
    k<SHAPE, FEED><<<grid, 256>>>(src, d, iters, 1.f, 2.f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s, e);
    const double tf = (double)grid * 4 * iters * 16 * 4096.0 / ms / 1e9;
    std::printf("%d waves/SIMD  %-40s rep %d: %7.1f ms  %6.1f TFLOP/s\n", wg_per_cu, name, rep, ms, tf);
  }
}

int main() {
  float *src, *d;
  (void)hipMalloc(&src, 1 << 26);
  (void)hipMemset(src, 0, 1 << 26);
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {4, 5}) {
    run<32, false>("32x32x2, registers only", w, src, d);
    run<16, false>("16x16x4, registers only", w, src, d);
    run<32, true>("32x32x2 + the loop's feed", w, src, d);
    run<16, true>("16x16x4 + the loop's feed", w, src, d);
  }
  return 0;
}
