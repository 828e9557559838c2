#!/usr/bin/env python3
"""Developer tool (GPU): throughput of the analysis front-end between HiFi-GAN and SiFiGAN and of the
SiFiGAN generator behind it, for B utterances of T mel frames (T * 240 samples each; 2 T + 1 analysis frames at 5 ms).

Per stage: HIP-event time, analysis frames/s, x real time, and for the two LDS-resident kernels the arithmetic they
do per frame against the chip's peaks -- they are bound by LDS traffic and barrier latency, not by HBM:
  CheapTrick: 3 real 1024-point transforms, each a 512-point complex FFT (256 * 9 butterflies * 10 flop, two radix-2
              stages per LDS pass) + a separation pass = 0.08 MFLOP and 3 * 6 passes * 512 points * 32 B = 0.29 MB of LDS
              traffic;
  D4C:        5 real + 2 complex 2048-point transforms = 5 * 512 * 10 * 10 + 2 * 1024 * 11 * 10 = 0.48 MFLOP and
              (5 * 6 * 1024 + 2 * 6 * 2048) * 32 B = 1.7 MB of LDS traffic, plus 3 bitonic sorts of 1024 doubles (mostly
              in registers) per voiced frame.
HBM traffic per frame is 120 new samples in (8 B each, the rest of the window hits L2) and 513 + 3 doubles out.
Prints one JSON object.  (Values are checked against oracle/world_oracle.py in tests/test_world.py.)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _shapes, sifigan, world  # noqa: E402
from serenade_amd.utils.synth import fill_state_dict  # noqa: E402

FS = 24000


def singing(n, seed):
    """harmonic source with a glide and vibrato through three resonances + noise; unvoiced stretches"""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    f = 220.0 * 2 ** (0.5 * np.sin(2 * np.pi * 0.13 * t + seed)) * (1 + 0.02 * np.sin(2 * np.pi * 5.5 * t))
    voiced = np.sin(2 * np.pi * 0.4 * t + seed) > -0.8
    ph = 2 * np.pi * np.cumsum(f) / FS
    src = sum(np.clip((10000.0 - k * f) / 1000.0, 0.0, 1.0) * np.cos(k * ph) for k in range(1, 40))
    a = np.array([1.0])
    for fc, bw in ((700, 130), (1800, 200), (3200, 300)):
        r = np.exp(-np.pi * bw / FS)
        a = np.convolve(a, [1.0, -2 * r * np.cos(2 * np.pi * fc / FS), r * r])
    x = lfilter([1.0], a, np.where(voiced, src, 0.0))
    x = 0.5 * x / np.abs(x).max() + (0.0005 + 0.02 * ~voiced) * rng.standard_normal(n)
    idx = np.minimum(np.arange(n // 240) * 240, n - 1)          # the decode CLI's 10 ms contour
    return x.astype(np.float32), np.where(voiced[idx], f[idx], 0.0)


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        out = fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, out


def run(B=8, T=1024, dev=None, with_generator=True):
    """the numbers as a dict (bench.py embeds them as the `analysis_stage` key of its line)"""
    n = T * 240
    dev = torch.device("cuda:0") if dev is None else dev
    sig = [singing(n, s) for s in range(B)]
    wave = torch.from_numpy(np.stack([w for w, _ in sig])).to(dev)
    lf0 = [f for _, f in sig]
    an = world.Analyzer()
    F = world.harvest_frame_count(n, FS)
    frames = B * F
    out = {"workload": f"B={B} x T={T} mel frames: {n / FS:.2f} s of audio each, {F} analysis frames at 5 ms",
           "frames": frames}
    # the stage as the pipeline runs it
    ms, (in_signal, c, dfs, feats) = timed(lambda: an(wave, [n] * B, lf0))
    out["analyzer_total"] = {"ms": ms, "frames_per_s": frames / ms * 1e3, "x_realtime": B * n / FS / ms * 1e3}
    voiced = int((feats["bap"][..., 0] != feats["bap"].max()).sum())
    out["frames_d4c_analysed"] = voiced
    # its kernels, separately
    x = torch.empty(wave.shape, dtype=torch.float64, device=dev)
    world.check(world._lib.lib().srn_wave_to_f64(wave.data_ptr(), x.data_ptr(), wave.numel(), 1, world._stream()), "w")
    f0 = feats["f0"]
    t = torch.from_numpy(np.arange(F) * 5.0 / 1000.0).to(dev)[None].expand(B, F).contiguous()
    x_len, nf = world._i32([n] * B, dev), feats["nf"]
    ms, (_, ceps) = timed(lambda: world._cheaptrick_raw(x, x_len, f0, t, nf, FS, -0.15, 71.0, 1024, False, True))
    out["cheaptrick"] = {"ms": ms, "us_per_frame_per_cu": ms * 1e3 / frames * 256, "fp64_tflops": 0.08e6 * frames / ms / 1e9,
                         "lds_tb_s": 0.29e6 * frames / ms / 1e9, "frames_per_s": frames / ms * 1e3}
    m = world._sp2mc_matrix(dev, 513, 39, 0.466, True)
    ms, _ = timed(lambda: world._project(ceps, m, False))
    out["sp2mc_projection"] = {"ms": ms, "fp64_tflops": 2 * 513 * 40 * frames / ms / 1e9}
    ms, _ = timed(lambda: world._d4c_raw(x, x_len, f0, t, nf, FS, 0.85))
    out["d4c"] = {"ms": ms, "us_per_voiced_frame_per_cu": ms * 1e3 / max(voiced, 1) * 256,
                  "fp64_tflops": 0.48e6 * voiced / ms / 1e9, "lds_tb_s": 1.7e6 * voiced / ms / 1e9,
                  "frames_per_s": frames / ms * 1e3}
    ms, _ = timed(lambda: an.excitation(feats))
    out["excitation_with_noise_draw"] = {"ms": ms}
    out["peaks"] = {"fp64_vector_tflops": 78.6, "lds_read_tb_s": 256 * 256 * 2.4e9 / 1e12,
                    "note": "one workgroup per frame (256 / 512 threads); bound by LDS round trips between barriers"}
    if not with_generator:
        return out
    # the generator behind it (a9), B = 1 as the reference's loop runs it, and the batch
    g = sifigan.SiFiGANGenerator(**sifigan.DEFAULT_PARAMS)
    g.load_state_dict(fill_state_dict(_shapes.as_meta(sifigan.sifigan_shapes(**sifigan.DEFAULT_PARAMS)), seed=1))
    g.remove_weight_norm()
    g = g.eval().to(dev)
    ms, _ = timed(lambda: g(in_signal, c, dfs), iters=3, warm=1)
    out["sifigan_generator"] = {"ms": ms, "frames_per_s": frames / ms * 1e3, "x_realtime": B * n / FS / ms * 1e3}
    return out


def main():
    B, T = (int(v) for v in (sys.argv[1:3] + ["8", "1024"][len(sys.argv) - 1:]))
    print(json.dumps(run(B, T), indent=1))


if __name__ == "__main__":
    main()
