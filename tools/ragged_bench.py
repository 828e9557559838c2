#!/usr/bin/env python3
"""Developer tool (GPU): the decode CLI's style loop (B = 1 per style) against one exact ragged batch of the same items."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serenade_amd  # noqa: E402
from serenade_amd import models  # noqa: E402
from serenade_amd.utils.synth import SERENADE_PARAMS, fill_state_dict, synth_inputs  # noqa: E402
from serenade_amd import _shapes  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = models.Serenade(**SERENADE_PARAMS)
    m.load_state_dict(fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**SERENADE_PARAMS)), seed=0))
    m = m.eval().to(dev)
    T = int(os.environ.get("SRN_T", "256"))
    refs = [int(v) for v in os.environ.get("SRN_REFS", "200,256,310,280").split(",")]
    ds = [synth_inputs(1, T, T_ref=r, seed=i) for i, r in enumerate(refs)]
    g = lambda d, k: d[k].to(dev)
    items = [tuple(g(d, k)[0] for k in ("x", "midi", "lft", "ref_x", "ref_logmel", "ref_midi", "ref_lft")) for d in ds]
    zs = [g(d, "z") for d in ds]
    for mode in ("fp32", "bf16x6", "bf16x3"):
        serenade_amd.set_precision(mode)

        def loop():
            return [m.inference(g(d, "x"), d["lengths"], g(d, "midi"), g(d, "lft"), g(d, "ref_x"), d["ref_lengths"],
                                g(d, "ref_logmel"), g(d, "ref_midi"), g(d, "ref_lft"), noise=z) for d, z in zip(ds, zs)]

        def batch():
            return m.inference_ragged(items, noises=[z[0] for z in zs])

        res = {}
        for name, fn in (("loop", loop), ("ragged", batch)):
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                out = fn()
            torch.cuda.synchronize()
            res[name] = ((time.perf_counter() - t0) / 10 * 1e3, out)
        err = max(((a - b).abs().max() / b.abs().max()).item() for a, b in zip(res["ragged"][1], res["loop"][1]))
        print(f"{mode}: T={T} refs={refs}: loop {res['loop'][0]:.2f} ms, ragged {res['ragged'][0]:.2f} ms, "
              f"max rel diff {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
