// mfma_f32_peak.hip -- stand-alone: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
//
// What does v_mfma_f32_32x32x2_f32 sustain from registers alone (no LDS, no memory), as a function of waves per SIMD
// and of independent accumulator chains per wave?  The product's exact-fp32 tiles give every wave ONE 32 x 32
// accumulator (a dependent chain) and rely on 4-6 co-resident waves per SIMD; this is the ceiling of that arrangement
// and the clock the card holds under it -- the number the kernels' 0.80-0.83 MFMA-busy should be read against.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int wg_per_cu, float* d) {
  const int cus = 256, iters = 4000;
  const int grid = cus * wg_per_cu;
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  chain_kernel<NACC><<<grid, 256>>>(d, 200, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  chain_kernel<NACC><<<grid, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0.f;
  hipEventElapsedTime(&ms, s, e);
  const double fl = (double)grid * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 2;
  // cycles per MFMA per SIMD if the pipe were never idle = 64: the implied clock at 100 % busy
  std::printf("waves/SIMD %d  chains/wave %d : %7.1f TFLOP/s  (%.3f of 157.3)  %.2f ms\n", wg_per_cu, NACC,
              fl / ms / 1e9, fl / ms / 1e9 / 157.3, ms);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 3, 4, 6, 8}) {
    run<1>(w, d);
    run<2>(w, d);
    run<4>(w, d);
  }
  return 0;
}
