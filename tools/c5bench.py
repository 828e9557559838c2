#!/usr/bin/env python3
"""Developer tool (GPU): BASELINE configs[4]'s per-GPU shape and its batch-8 variant, B x T=4096 (L = 4352), 10 Euler
steps + HiFi-GAN, in every arithmetic mode and with the attention-only modes (serenade_amd.set_attention_precision).
Prints one JSON object: ms per batch, frames/s, TFLOP/s of SURVEY 8(d)'s algorithmic work."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import serenade_amd  # noqa: E402
from serenade_amd.utils.synth import synth_inputs  # noqa: E402


def main():
    B, T = (int(v) for v in (sys.argv[1:3] + ["8", "4096"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    model, voc, _, _ = bench.build_models(dev)
    d = synth_inputs(B, T, T_ref=256, seed=1235)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}

    def step():
        mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                              g["ref_midi"], g["ref_lft"], n_timesteps=10, noise=g["z"])
        return mel, voc.decode_batch(mel if mel.dim() == 3 else mel.unsqueeze(0))

    out = {"workload": f"B={B} x T={T} (T_ref 256), 10 Euler steps + HiFi-GAN"}
    ref = None
    for main_p, attn in (("fp32", None), ("fp32", "bf16x6"), ("fp32", "bf16x3"), ("bf16x6", None), ("bf16x3", None)):
        serenade_amd.set_precision(main_p)
        serenade_amd.set_attention_precision(attn)
        mel, wave = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            mel, wave = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        if ref is None:
            ref = (mel.clone(), wave.clone())
        key = main_p + (f"+attention_{attn}" if attn else "")
        out[key] = {"ms_per_batch": dt * 1e3, "frames_per_s": B * T / dt,
                    "tflops": bench.algorithmic_flops(B, T, 256, 10) / dt / 1e12,
                    "mel_rel_to_fp32": float((mel - ref[0]).abs().max() / ref[0].abs().max()),
                    "wave_abs_to_fp32": float((wave - ref[1]).abs().max())}
    serenade_amd.set_attention_precision(None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
