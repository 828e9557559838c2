// mfma_valu_mix.hip -- stand-alone: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip && ./mfma_valu_mix
//
// The fp32 matrix pipe (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD) and the fp32 vector pipe (v_fma_f32, 64 FLOP/clk/SIMD)
// are separate pipes of one SIMD.  How many exact-fp32 FLOP/s does a register-only loop sustain when a wave issues NV
// v_fma_f32 (or v_pk_fma_f32) between two MFMAs -- i.e. is a contraction kernel that gives part of its tile to the
// vector pipe worth building, or does the issue port / the power cap hand the gain back?
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool PK>
__global__ __launch_bounds__(256) void mix_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  constexpr int NC = NV > 0 ? NV : 1;
  float c[NC];
  f32x2 c2[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    c[j] = 0.f;
    c2[j] = {0.f, 0.f};
  }
  f32x2 a2 = {a, a + 1.f}, b2 = {b, b - 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if constexpr (PK)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c2[j]) : "v"(a2), "v"(b2));
        else
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += acc[e];
#pragma unroll
  for (int j = 0; j < NC; ++j) s += c[j] + c2[j].x + c2[j].y;
  if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool PK>
void run(int wg_per_cu, float* d) {
  const int cus = 256, iters = 3000;
  const int grid = cus * wg_per_cu;
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  mix_kernel<NV, PK><<<grid, 256>>>(d, 300, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(s);
  mix_kernel<NV, PK><<<grid, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0.f;
  hipEventElapsedTime(&ms, s, e);
  const double n_mfma = (double)grid * 4 * iters * 16;
  const double fl_m = n_mfma * 2.0 * 32 * 32 * 2;
  const double fl_v = n_mfma * NV * 64 * 2.0 * (PK ? 2 : 1);
  std::printf("waves/SIMD %d  %s x %2d per MFMA : MFMA %6.1f + VALU %6.1f = %6.1f TFLOP/s (%.3f of 157.3)  %.2f ms\n",
              wg_per_cu, PK ? "v_pk_fma" : "v_fma   ", NV, fl_m / ms / 1e9, fl_v / ms / 1e9, (fl_m + fl_v) / ms / 1e9,
              (fl_m + fl_v) / ms / 1e9 / 157.3, ms);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4, 6}) {
    run<0, false>(w, d);
    run<2, false>(w, d);
    run<4, false>(w, d);
    run<8, false>(w, d);
    run<12, false>(w, d);
    run<16, false>(w, d);
    run<24, false>(w, d);
    run<2, true>(w, d);
    run<4, true>(w, d);
    run<8, true>(w, d);
    run<12, true>(w, d);
  }
  return 0;
}
