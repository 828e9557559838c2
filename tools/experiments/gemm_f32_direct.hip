// gemm_f32_direct.hip -- stand-alone: hipcc --offload-arch=gfx950 -O3 -o gemm_f32_direct gemm_f32_direct.hip && ./gemm_f32_direct
//
// Question: the product's exact-fp32 tiles stage both operands through LDS behind two barriers per 32-deep step, and
// every re-arrangement of that feed measured flat or slower.  What does the SAME 32 x 32-per-wave
// MFMA schedule reach with NO LDS and NO barrier -- every wave fetching its own fragments straight from global memory
// (L1 / L2 hits for the rows its three sister waves of the 64 x 64 tile also read), double-buffered in registers?
// C[m][n] = sum_k A[m][k] B[n][k], both row-major (the layout of srn_conv_gemm's 1-tap case); checked against fp64 on
// a sample of entries.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Frag {
  float4 a[4], b[4];
};

__device__ __forceinline__ void load_frag(Frag& f, const float* __restrict__ pa, const float* __restrict__ pb) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    f.a[kk] = *reinterpret_cast<const float4*>(pa + 8 * kk);
    f.b[kk] = *reinterpret_cast<const float4*>(pb + 8 * kk);
  }
}

__device__ __forceinline__ void mma_frag(f32x16& acc, const Frag& f) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk].x, f.b[kk].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk].y, f.b[kk].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk].z, f.b[kk].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk].w, f.b[kk].w, acc, 0, 0, 0);
  }
}

// 256 threads: four independent waves, wave w owns the 32 x 32 quadrant (w >> 1, w & 1) of a 64 x 64 tile
__global__ __launch_bounds__(256) void gemm_direct(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int mt = blockIdx.x / n_tiles, nt = blockIdx.x - mt * n_tiles;
  const int m0 = mt * 64 + (wave >> 1) * 32, n0 = nt * 64 + (wave & 1) * 32;
  const float* pa = A + (int64_t)(m0 + li) * K + 4 * lh;
  const float* pb = B + (int64_t)(n0 + li) * K + 4 * lh;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  Frag f0, f1;
  load_frag(f0, pa, pb);
  int k = 0;
  for (; k + 64 <= K; k += 64) {  // two 32-deep steps per trip: static register names for the double buffer
    load_frag(f1, pa + k + 32, pb + k + 32);
    mma_frag(acc, f0);
    if (k + 64 < K) load_frag(f0, pa + k + 64, pb + k + 64);
    mma_frag(acc, f1);
  }
  if (k < K) mma_frag(acc, f0);  // K % 64 == 32
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    C[(int64_t)row * N + n0 + li] = acc[e];
  }
}

static void run(int M, int N, int K) {
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  float *A, *B, *C;
  hipMalloc(&A, hA.size() * 4);
  hipMalloc(&B, hB.size() * 4);
  hipMalloc(&C, (size_t)M * N * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const int n_tiles = N / 64, grid = (M / 64) * n_tiles;
  for (int i = 0; i < 3; ++i) gemm_direct<<<grid, 256>>>(A, B, C, M, N, K, n_tiles);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  hipEventRecord(s);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) gemm_direct<<<grid, 256>>>(A, B, C, M, N, K, n_tiles);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0.f;
  hipEventElapsedTime(&ms, s, e);
  ms /= reps;
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int t = 0; t < 200; ++t) {
    const int m = rand() % M, n = rand() % N;
    double r = 0.0;
    for (int k = 0; k < K; ++k) r += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
    worst = fmax(worst, fabs(r - hC[(size_t)m * N + n]));
  }
  std::printf("M %6d N %5d K %5d : %8.1f us  %6.1f TFLOP/s  (max |err| on 200 entries %.2e)\n", M, N, K, ms * 1e3,
              2.0 * M * N * K / ms / 1e9, worst);
  hipFree(A);
  hipFree(B);
  hipFree(C);
}

int main() {
  run(10240, 512, 1536);   // a k = 3 conv of the UNet at L = 1280, B = 8
  run(10240, 512, 2048);
  run(10240, 6144, 512);   // the fused q | k | v projection
  run(5120, 512, 1536);    // the half-resolution level
  return 0;
}
