#!/usr/bin/env python3
"""Developer tool (GPU): throughput of the SiFiGAN V2 generator (row a9, parity unpinned) at the headline audio
length: B utterances x 2048 5-ms frames (10.24 s of 24 kHz audio each), full-width config."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _shapes, ops, sifigan  # noqa: E402
from serenade_amd.utils.synth import fill_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    cfg = sifigan.DEFAULT_PARAMS
    dev = torch.device("cuda:0")
    g = sifigan.SiFiGANGenerator(**cfg)
    g.load_state_dict(fill_state_dict(_shapes.as_meta(sifigan.sifigan_shapes(**cfg)), seed=2))
    g = g.eval().to(dev)
    rng = np.random.default_rng(0)
    hop = int(np.prod(cfg["upsample_scales"]))
    c = torch.from_numpy(rng.standard_normal((B, cfg["in_channels"], T)).astype(np.float32)).to(dev)
    f0 = rng.uniform(100, 400, (B, 1, T))
    x = torch.from_numpy((0.1 * np.sin(np.cumsum(np.repeat(2 * np.pi * f0 / 24000, hop, axis=2), axis=2)))
                         .astype(np.float32)).to(dev)
    d = [torch.from_numpy(np.repeat(24000.0 / f0 / df / 16.0, us, axis=2).astype(np.float32)).to(dev)
         for df, us in zip((0.5, 1, 4, 8), np.cumprod(cfg["upsample_scales"]))]
    for _ in range(2):
        g(x, c, d)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        g(x, c, d)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    sec = B * T * hop / 24000.0
    print(f"SiFiGAN B={B} T={T} (5 ms frames): {el * 1e3:.2f} ms per batch = {sec / el:.0f}x real time, "
          f"{B * T / 2 / el:.0f} mel-frame equivalents (10 ms) per second")
    if "--ops" in sys.argv:
        # per-op table (the a8 evidence of tools/opbench.py for row a9): every contraction by shape with its TFLOP/s, the
        # HBM-bound gathers with their GB/s, and the generator's roofline line
        import collections
        import json
        pl = list(g._plans.values())[0] if hasattr(g, "_plans") and g._plans else None
        if pl is not None:
            def t_op(op, reps=5):
                op()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    op()
                e.record()
                torch.cuda.synchronize()
                return s.elapsed_time(e) / reps
            agg, other = collections.OrderedDict(), collections.OrderedDict()
            for op in pl.ops:
                ms = t_op(op)
                if isinstance(op, ops.ConvOp):
                    k = op.kw
                    key = (k["n_batch"], k["T_out"], k["N"], len(k.get("taps", (0,))) * k["C_in"], k.get("in_stride", 1),
                           int(k.get("pro_act", 0)))
                    a = agg.setdefault(key, [0, 0.0])
                else:
                    a = other.setdefault(type(op).__name__ + ":" + getattr(op, "name", ""), [0, 0.0])
                a[0] += 1
                a[1] += ms
            tot = sum(v[1] for v in agg.values())
            fl = sum(2.0 * k[0] * k[1] * k[2] * k[3] * c for k, (c, _) in agg.items())
            print(f"contractions: {tot:.2f} ms for {fl / 1e9:.0f} GFLOP = {fl / tot / 1e9:.1f} TFLOP/s; other kernels "
                  f"{sum(v[1] for v in other.values()):.2f} ms")
            print("   Z   T_out     N      K str act | cnt  ms_each   TF/s  share")
            for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                Z, To, N, K, st, act = key
                print(f"{Z:4d} {To:7d} {N:5d} {K:6d} {st:3d} {act:3d} | {cnt:3d} {ms / cnt:8.3f} "
                      f"{2.0 * Z * To * N * K * cnt / ms / 1e9:6.1f} {100 * ms / tot:6.1f}%")
            for name, (cnt, ms) in sorted(other.items(), key=lambda kv: -kv[1][1]):
                print(f"  {name:40s} cnt {cnt:3d}  {ms:7.3f} ms")
            peak = 157.3 if os.environ.get("SERENADE_AMD_PRECISION", "fp32") == "fp32" else 2500.0
            print(json.dumps({"sifigan_generator": {"ms_per_batch": el * 1e3, "contraction_ms": tot, "algorithmic_gflop": fl / 1e9,
                                                    "roofline": {"bound": "mfma", "achieved": fl / tot / 1e9, "peak": peak,
                                                                 "unit": "TFLOP/s", "frac": fl / tot / 1e9 / peak},
                                                    "precision": os.environ.get("SERENADE_AMD_PRECISION", "fp32")}}))


if __name__ == "__main__":
    main()
