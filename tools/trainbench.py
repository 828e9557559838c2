#!/usr/bin/env python3
"""Developer tool (GPU): the estimator training step of bench.py on its own (for rocprofv3 runs).

    python3 tools/trainbench.py [B] [L] [steps]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import serenade_amd  # noqa: E402


def main():
    B, L, steps = (int(v) for v in (sys.argv[1:4] + ["4", "1024", "5"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    _, _, sd, _ = bench.build_models(dev)
    serenade_amd.set_precision(os.environ.get("SERENADE_AMD_PRECISION", "fp32"))
    print(json.dumps(bench.train_step_bench(dev, sd, B=B, L=L, steps=steps), indent=1))


if __name__ == "__main__":
    main()
