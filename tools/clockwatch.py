#!/usr/bin/env python3
"""Developer tool (GPU): sample rocm-smi clocks / power while a long GEMM loop runs, or (--convert B T) while whole
conversions of B utterances x T frames (10 Euler steps + HiFi-GAN, exact fp32) run back to back."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, N, K = (int(os.environ.get(k, d)) for k, d in (("SRN_M", 10240), ("SRN_N", 2048), ("SRN_K", 2048)))
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    prec = (_lib.PREC_FP32 if "--fp32" in sys.argv else
            (_lib.PREC_BF16X6 if "--bf16x6" in sys.argv else _lib.PREC_BF16X3))
    op = ops.ConvOp(in0=x, w=w, out=out, n_batch=1, T_in=M, T_out=M, C_in=K, N=N, ld_in0=K, ldw=K, ld_out=N,
                    precision=prec, tile=int(os.environ.get("SRN_TILE", "1")))
    label = None
    if "--convert" in sys.argv:
        import bench
        import serenade_amd
        from serenade_amd.utils.synth import synth_inputs
        i = sys.argv.index("--convert")
        B, T = int(sys.argv[i + 1]), int(sys.argv[i + 2])
        serenade_amd.set_precision("fp32")
        model, voc, _, _ = bench.build_models(dev)
        g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in synth_inputs(B, T, T_ref=256, seed=1235).items()}

        def op():
            mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                                  g["ref_midi"], g["ref_lft"], noise=g["z"])
            voc.decode_batch(mel if mel.dim() == 3 else mel.unsqueeze(0))
        for _ in range(3):
            op()
        torch.cuda.synchronize()
        label = f"conversions B={B} x T={T}, 10 Euler steps + HiFi-GAN, exact fp32"
    stop = False
    seen = {"sclk": [], "power": []}

    def watch():
        import re
        while not stop:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
            for ln in r.stdout.splitlines():
                if "sclk" in ln or "Power" in ln or "mclk" in ln:
                    print(ln.strip(), flush=True)
                m = re.search(r"sclk clock level.*\((\d+)Mhz\)", ln)
                if m:
                    seen["sclk"].append(int(m.group(1)))
                m = re.search(r"Power \(W\): ([0-9.]+)", ln)
                if m:
                    seen["power"].append(float(m.group(1)))
            time.sleep(0.5)

    th = threading.Thread(target=watch)
    th.start()
    t0 = time.time()
    n = 0
    per = 20 if label else 200
    while time.time() - t0 < 4.0:
        for _ in range(per):
            op()
        torch.cuda.synchronize()
        n += per
    el = time.time() - t0
    stop = True
    th.join()
    import statistics
    med = lambda v: statistics.median(v[1:]) if len(v) > 1 else (v[0] if v else float("nan"))
    name = {_lib.PREC_FP32: "fp32", _lib.PREC_BF16X3: "bf16x3", _lib.PREC_BF16X6: "bf16x6"}[prec]
    if label:
        print(f"SUMMARY {label}: {n} in {el:.2f} s -> {el / n * 1e3:.2f} ms each; median sclk {med(seen['sclk'])} MHz, "
              f"median power {med(seen['power'])} W")
        return
    print(f"SUMMARY {name} {M}x{N}x{K}: {n} launches in {el:.2f} s -> {el / n * 1e6:.1f} us each, "
          f"{2.0 * M * N * K * n / el / 1e12:.1f} TF/s; median sclk {med(seen['sclk'])} MHz, median power "
          f"{med(seen['power'])} W")


if __name__ == "__main__":
    main()
