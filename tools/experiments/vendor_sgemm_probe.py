#!/usr/bin/env python3
"""Experiment (GPU): what the vendor library (rocBLAS / hipBLASLt behind torch.matmul, exact fp32) reaches on the GEMM
shapes behind the headline's convs -- a reference point for the 0.78-0.83 of the fp32 MFMA peak the own tiles reach.
C (M, N) = A (M, K) @ W (N, K)^T.  Not used by the product."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for M, N, K in ((10240, 512, 1536), (10240, 512, 2048), (10240, 6144, 512), (10240, 4096, 512), (5120, 512, 1536),
                (34816, 512, 1536)):
    a, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        c = a @ w.t()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"M {M:6d} N {N:5d} K {K:5d}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:6.1f} TFLOP/s")
