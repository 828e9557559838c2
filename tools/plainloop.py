#!/usr/bin/env python3
"""Developer tool (GPU): the headline workload's step loop with no per-launch events (for kernel traces / gap analysis)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from serenade_amd.utils.synth import synth_inputs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    model, voc, sd, gsd = bench.build_models(dev)
    d = synth_inputs(bench.B_PER_GPU, bench.T_SRC, T_ref=bench.T_REF, seed=1235)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}

    def step():
        mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"],
                              g["ref_logmel"], g["ref_midi"], g["ref_lft"], n_timesteps=bench.N_EULER, noise=g["z"])
        return voc.decode_batch(mel)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{el / n * 1e3:.2f} ms/step, {bench.B_PER_GPU * bench.T_SRC * n / el:.0f} frames/s")


if __name__ == "__main__":
    main()
