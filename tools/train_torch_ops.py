#!/usr/bin/env python3
"""Developer tool (GPU): which lines of serenade_amd/training.py launch torch's own kernels (copies, adds, fills ...)
during one eager training step, and what they cost.  torch.profiler with stacks; own kernels (srn_* calls through
ctypes) do not show up as aten ops and are left out on purpose.

    python3 tools/train_torch_ops.py [B] [L]
"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import serenade_amd  # noqa: E402
from serenade_amd import training  # noqa: E402


def main():
    B, L = (int(v) for v in (sys.argv[1:3] + ["4", "1024"][len(sys.argv) - 1:]))
    dev = torch.device("cuda:0")
    _, _, sd, _ = bench.build_models(dev)
    serenade_amd.set_precision("fp32")
    model = training.TrainSerenade(sd, dev, dropout=0.05)
    opt, sync = training.AdamW(model), training.GradSync(model)
    g = torch.Generator().manual_seed(4321)
    lens = torch.tensor([L - 37 * i for i in range(B)]).to(dev)
    x, logmel = torch.randn(B, L, 768, generator=g).to(dev), torch.randn(B, L, 80, generator=g).to(dev)
    midi, lft = torch.randn(B, L, 1, generator=g).to(dev), torch.randn(B, L, 1, generator=g).to(dev)

    def step():
        ret = model(x, lens, logmel, midi, lft)
        model.backward(ret["cfm_loss"] + ret["prior_loss"], sync)
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    by_line = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
    total = 0.0
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0.0) or 0.0
        if t <= 0 or not ev.name.startswith("aten::"):
            continue
        where = "?"
        for fr in ev.stack or ():
            if "serenade_amd/training.py" in fr:
                where = fr.split("serenade_amd/")[-1]
                break
        if where == "?":  # no python stacks on this stack: the op and its operand shapes identify the call site
            where = f"{ev.name} {[tuple(s) for s in (ev.input_shapes or []) if s][:3]}"
        r = by_line[where]
        r[0] += t
        r[1] += 1
        r[2][ev.name] += 1
        total += t
    print(f"# torch-launched kernel time in one eager step, B={B} x L={L}: {total / 1e3:.3f} ms")
    for where, (t, n, names) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{t:9.1f} us {n:4d}  {where:70s} {dict(names.most_common(3))}")


if __name__ == "__main__":
    main()
