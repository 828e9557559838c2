#!/usr/bin/env python3
"""Developer tool (GPU): one srn_conv_gemm shape, launched repeatedly (for PC sampling / counter runs on a single kernel).

    python3 tools/oneop.py [B] [T] [C_in] [N] [taps] [reps]     # exact fp32, channels-last, k-major weights
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _lib, ops  # noqa: E402


def main():
    B, T, C, N, k, reps = (int(v) for v in (sys.argv[1:7] + ["8", "1280", "512", "512", "3", "200"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    x, w = torch.randn(B, T, C, device=dev), torch.randn(N, k * C, device=dev) * 0.02
    y = torch.empty(B, T, N, device=dev)
    op = ops.ConvOp(in0=x, w=w, out=y, n_batch=B, T_in=T, T_out=T, C_in=C, N=N, in0_bs=T * C, ld_in0=C, ldw=k * C,
                    out_bs=T * N, ld_out=N, taps=ops.conv_taps(k), precision=_lib.PREC_FP32)
    for _ in range(5):
        op()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        op()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"B={B} T={T} C_in={C} N={N} taps={k}: {ms * 1e3:.1f} us, {2.0 * B * T * N * k * C / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
