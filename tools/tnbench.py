#!/usr/bin/env python3
"""Developer tool (GPU): srn_tn_gemm on the training step's shapes (B = 4 x L = 1024 unless given): HIP-event time and
TFLOP/s per shape.  Prints one JSON object."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd.ops import TnGemmOp  # noqa: E402


def timed(op, iters=20):
    for _ in range(3):
        op()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        op()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B, L = (int(v) for v in (sys.argv[1:3] + ["4", "1024"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    r = lambda *s: torch.randn(*s, device=dev)
    out = {"workload": f"B={B} x L={L}"}
    shapes = {"conv k3 512<-512": (512, 512, (-1, 0, 1), L), "conv k3 512<-1024": (512, 1024, (-1, 0, 1), L // 2),
              "linear qkv 6144<-512": (6144, 512, (0,), L), "linear geglu 4096<-512": (4096, 512, (0,), L),
              "linear o 512<-2048": (512, 2048, (0,), L), "conv 1x1 512<-256": (512, 256, (0,), L),
              "final_proj 80<-512": (80, 512, (0,), L)}
    tot_ms = tot_fl = 0.0
    for name, (N, C, taps, T) in shapes.items():
        x, dy = r(B, T, C), r(B, T, N)
        dw = torch.empty(N, len(taps) * C, device=dev)
        op = TnGemmOp(a=dy, b=x, out=dw, n_items=B, T_a=T, T_b=T, M=N, N=C, lda=N, ldb=C, ldc=len(taps) * C, shifts=taps,
                      a_is=T * N, b_is=T * C)
        ms = timed(op)
        fl = 2.0 * B * T * N * C * len(taps)
        out[name] = {"us": ms * 1e3, "tflops": fl / ms / 1e9, "sliced": op._ws is not None}
        tot_ms, tot_fl = tot_ms + ms, tot_fl + fl
    H, hd = 4, 512
    for Lq in (L, L // 2):
        Lp = (Lq + 31) // 32 * 32
        P, do = r(B, H, Lq, Lp), r(B, Lq, H * hd)
        o = torch.zeros(B, Lq, 3 * H * hd, device=dev)
        op = TnGemmOp(a=P, b=do, out=(o, 2 * H * hd), n_items=1, T_a=Lq, T_b=Lq, M=Lq, N=hd, lda=Lp, ldb=H * hd,
                      ldc=3 * H * hd, n_batch=B, n_head=H, a_bs=H * Lq * Lp, a_hs=Lq * Lp, b_bs=Lq * H * hd, b_hs=hd,
                      out_bs=Lq * 3 * H * hd, out_hs=hd)
        ms = timed(op)
        fl = 2.0 * B * H * Lq * Lq * hd
        out[f"attention dV L={Lq}"] = {"us": ms * 1e3, "tflops": fl / ms / 1e9}
        tot_ms, tot_fl = tot_ms + ms, tot_fl + fl
    out["all"] = {"ms": tot_ms, "tflops": tot_fl / tot_ms / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
