#!/usr/bin/env python3
"""Developer experiment (GPU): accuracy and speed of the split-bf16 (bf16x3) contraction mode vs exact fp32 MFMA."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from serenade_amd import _lib, ops  # noqa: E402
from serenade_amd.utils.synth import synth_inputs  # noqa: E402


def run(model, voc, g, reps=3):
    mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                          g["ref_midi"], g["ref_lft"], noise=g["z"])
    wave = voc.decode_batch(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m2 = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"],
                             g["ref_logmel"], g["ref_midi"], g["ref_lft"], noise=g["z"])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        voc.decode_batch(m2)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return mel, wave, dt


def main():
    dev = torch.device("cuda:0")
    B, T, Tr = 8, 1024, 256
    model, voc, sd, gsd = bench.build_models(dev)
    d = synth_inputs(B, T, T_ref=Tr, seed=1235)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}
    res = {}
    for name, prec in (("fp32", _lib.PREC_FP32), ("bf16x3", _lib.PREC_BF16X3)):
        ops.DEFAULT_PRECISION = prec
        model._invalidate()
        voc.model._invalidate()
        mel, wave, dt = run(model, voc, g)
        res[name] = (mel.clone(), wave.clone(), dt)
        print(f"{name}: {dt * 1e3:.1f} ms per batch -> {B * T / dt:.0f} frames/s")
    m0, w0, _ = res["fp32"]
    m1, w1, _ = res["bf16x3"]
    print("bf16x3 vs fp32 (GPU): mel rel err", ((m1 - m0).abs().max() / m0.abs().max()).item(),
          " wave abs err", (w1 - w0).abs().max().item())
    # vocoder only, same mel
    ops.DEFAULT_PRECISION = _lib.PREC_BF16X3
    voc.model._invalidate()
    wb = voc.decode_batch(m0)
    print("vocoder-only bf16x3 vs fp32 on the same mel: wave abs err", (wb - w0).abs().max().item())


if __name__ == "__main__":
    main()
