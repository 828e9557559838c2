#!/usr/bin/env python3
"""Developer tool (GPU): throughput of the feature front-end -- log-mel (fft 512 / win 480 / hop 240 /
80 mels, conf/serenade.yaml:4-21) and A-weighted loudness (n_fft 2048) of B utterances of T frames.  Prints one JSON
object.  (The comparison with the numpy restatement, values and host timing, is tests/test_features.py: only tests may
use oracle/.)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import features  # noqa: E402


def main():
    B, T = (int(v) for v in (sys.argv[1:3] + ["8", "1024"][len(sys.argv) - 1:]))
    sr, hop = 24000, 240
    n = T * hop
    rng = np.random.default_rng(0)
    audio = (rng.standard_normal((B, n)) * 0.1).astype(np.float32)
    a = torch.from_numpy(audio).cuda()
    kw = dict(fft_size=512, hop_size=hop, win_length=480, num_mels=80, fmin=63, fmax=12000)
    out = {"workload": f"B={B} utterances x {T} frames ({n / sr:.2f} s each at 24 kHz)"}
    for name, fn in (("logmel", lambda: features.logmelfilterbank(a, sr, **kw)),
                     ("loudness", lambda: features.loudness_extract(a, sr, hop))):
        for _ in range(3):
            r = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        out[name] = {"ms": dt * 1e3, "frames_per_s": B * r.shape[1] / dt, "x_realtime": B * n / sr / dt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
