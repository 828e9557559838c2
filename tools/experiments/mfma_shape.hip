// Developer experiment (not part of the library): which bf16 MFMA shape holds the higher clock at the board's power cap?
// Two register-only loops on random operands, 4 waves per workgroup, 2 workgroups per CU, same FLOPs per iteration:
//   A: 4 x v_mfma_f32_32x32x16_bf16  (4 accumulator tiles of 32 x 32)
//   B: 16 x v_mfma_f32_16x16x32_bf16 (16 accumulator tiles of 16 x 16) -- hmm: same FLOPs = 8 of them; see below
// FLOPs: 32x32x16 = 32768 per instruction, 16x16x32 = 16384 -> 2 of B per 1 of A.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = src[(tid * 8 + i) & 0xffff];
    b[i] = src[(tid * 8 + 4 + i) & 0xffff];
  }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + 1) & 3], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 1) & 3], b[k], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 2) & 3], b[(k + 3) & 3], acc[3], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k + j) & 3], b[(k + (j >> 1)) & 3], acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[tid] = s;
}

int main() {
  const int blocks = 256 * 2, iters = 20000;
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // random bf16 of magnitude ~1
  bf16x8* src;
  float* out;
  hipMalloc(&src, h.size() * 2);
  hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int shape : {32, 16, 32, 16}) {
    const double flops = (double)blocks * 4 * iters * 4 * 4 * 32768.0;  // waves x iters x k x 4 (32x32x16) or 8 (16x16x32)
    float best = 1e30f;
    double total = 0;
    int reps = 0;
    const double t_end = 3.0;  // seconds per arm: long enough for the power management to settle
    while (total < t_end) {
      hipEventRecord(e0);
      if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      total += ms * 1e-3;
      ++reps;
      if (total > t_end * 0.5 && ms < best) best = ms;
    }
    printf("shape %dx%d: %d launches, last-half best %.3f ms -> %.0f TFLOP/s (bf16 MFMA only, registers)\n", shape, shape,
           reps, best, flops / (best * 1e-3) / 1e12);
    fflush(stdout);
  }
  return 0;
}
