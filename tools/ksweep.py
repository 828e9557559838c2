#!/usr/bin/env python3
"""Developer tool (GPU only): time one plain GEMM shape over a sweep of K and fit t = a + b*K, which separates the
per-block fixed cost (prologue + epilogue, a) from the main-loop rate (b).

    python tools/ksweep.py [--fp32] [M N]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _lib, ops  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    M, N = (int(args[0]), int(args[1])) if len(args) >= 2 else (10240, 2048)
    prec = _lib.PREC_FP32 if "--fp32" in sys.argv else _lib.PREC_BF16X3
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    ks = [128, 256, 512, 1024, 2048, 4096]
    ts = []
    out = torch.empty(M, N, device=dev)
    for K in ks:
        x = torch.randn(M, K, generator=g).to(dev)
        w = torch.randn(N, K, generator=g).to(dev)
        for tile in (0, 1, 2, 4, 6):
            op = ops.ConvOp(in0=x, w=w, out=out, n_batch=1, T_in=M, T_out=M, C_in=K, N=N, ld_in0=K, ldw=K, ld_out=N,
                            precision=prec, tile=tile)
            op()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            s.record()
            for _ in range(reps):
                op()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / reps
            if tile == 0:
                ts.append(ms)
            print(f"M={M} N={N} K={K:5d} tile={tile} {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
    b, a = np.polyfit(np.array(ks, dtype=np.float64), np.array(ts), 1)
    print(f"fit (tile auto): t = {a * 1e3:.1f} us + {b * 1e3 * 32:.3f} us per 32-deep k-step; asymptotic "
          f"{2.0 * M * N / b / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
