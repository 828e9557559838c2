#!/usr/bin/env python3
"""Probe (GPU): which ingredient of a captured training step goes wrong on replay on this ROCm / torch stack?
VERDICT r2 item 3 / ADVICE r2: `GraphedStep` had to (1) avoid per-leaf AccumulateGrad and (2) avoid torch's multi-block
reductions inside the capture.  Each experiment captures a tiny region with torch.cuda.graph, replays it three times on
changing inputs and compares every replay with the eager result.  Prints one JSON object; no conclusions in code."""
import ctypes
import json

import torch


def replays(build, feed, eager, n=3):
    """build() -> (graph, outputs getter); feed(i) writes inputs of replay i; eager(i) -> expected"""
    g, get = build()
    res = []
    for i in range(n):
        feed(i)
        exp = eager(i)
        g.replay()
        torch.cuda.synchronize()
        got = get()
        err = max(float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)) for a, b in zip(got, exp))
        res.append(err)
    return res


def exp_reduction(warm_side, n=1 << 24, op="sum"):
    dev = torch.device("cuda:0")
    x = torch.zeros(n, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(0)
    src = [torch.randn(n, generator=gen).to(dev) for _ in range(3)]
    f = {"sum": lambda t: t.sum(), "norm": lambda t: t.norm(), "colsum": lambda t: t.view(-1, 512).sum(0),
         "rowsum": lambda t: t.view(512, -1).sum(1), "mean_dim": lambda t: t.view(4, -1).mean(1)}[op]
    out = {}

    def build():
        if warm_side:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    f(x)
            torch.cuda.current_stream().wait_stream(s)
        else:
            for _ in range(2):
                f(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out["y"] = f(x)
        return g, lambda: [out["y"].clone()]

    return replays(build, lambda i: x.copy_(src[i]), lambda i: [f(src[i])])


def exp_memset():
    """raw hipMemsetAsync captured in a graph, followed by an increment: does the memset run on every replay?"""
    hip = ctypes.CDLL("libamdhip64.so")
    buf = torch.full((64,), 5, device="cuda:0", dtype=torch.int32)
    one = torch.ones(64, device="cuda:0", dtype=torch.int32)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(64 * 4), st)
        buf.add_(one)
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append(int(buf[0]))
    return {"rc": rc, "value_after_each_replay": vals, "expected": [1, 1, 1]}


def exp_memset_sweep():
    """sizes x values of hipMemsetAsync nodes in ONE capture, each followed by +1; three replays"""
    hip = ctypes.CDLL("libamdhip64.so")
    cases = [(n, v) for n in (4, 8, 64, 256, 1024, 4096, 1 << 16) for v in (0, 1, 255)]
    bufs = [torch.full((n,), 7, device="cuda:0", dtype=torch.uint8) for n, _ in cases]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for b, (n, v) in zip(bufs, cases):
            hip.hipMemsetAsync(ctypes.c_void_p(b.data_ptr()), v, ctypes.c_size_t(n), st)
            b.add_(1)
    out = {}
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        for b, (n, v) in zip(bufs, cases):
            exp = (v + 1) & 255
            ok = bool((b == exp).all())
            out.setdefault(f"{n}B_value{v}", []).append("ok" if ok else f"first_bytes={b[:8].tolist()}")
    return out


def exp_many_reductions():
    """forty torch reductions of training-step shapes in ONE capture (the allocator hands their semaphore buffers out
    of the graph's private pool back to back)"""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(4096, 512), (4096, 2048), (8192, 512), (2048, 1536), (1 << 22,), (84_000_000,), (4, 80, 1024), (4096, 4096)]
    xs = [torch.zeros(s, device=dev) for s in shapes]
    srcs = [[torch.randn(s, generator=gen).to(dev) for s in shapes] for _ in range(3)]

    def fns(x):
        r = [x.sum(), x.norm(), (x * x).sum()]
        if x.dim() >= 2:
            r += [x.sum(0), x.reshape(x.shape[0], -1).sum(1)]
        return r

    for x in xs:
        fns(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.graph(g):
        for x in xs:
            outs.append(fns(x))
    worst = []
    for i in range(3):
        for x, s in zip(xs, srcs[i]):
            x.copy_(s)
        g.replay()
        torch.cuda.synchronize()
        w = 0.0
        for x, o in zip(xs, outs):
            for a, b in zip(o, fns(x)):
                w = max(w, float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)))
        worst.append(w)
    return worst


def exp_graphed_step_nodes():
    """node kinds of the product's captured training step (GraphedStep) from hipGraphDebugDotPrint"""
    import os
    import re
    import sys
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from serenade_amd import _shapes, training
    from serenade_amd.utils.synth import SERENADE_PARAMS, fill_state_dict
    dev = torch.device("cuda:0")
    sd = fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**SERENADE_PARAMS)), seed=0)
    model = training.TrainSerenade(sd, dev, dropout=0.05)
    opt = training.AdamW(model)
    real = torch.cuda.CUDAGraph

    class Dbg(real):
        def __new__(cls, *a, **k):
            g = real.__new__(cls)
            return g

        def __init__(self, *a, **k):
            super().__init__()
            self.enable_debug_mode()

    torch.cuda.CUDAGraph = Dbg
    try:
        step = training.GraphedStep(model, opt, 2, 256)
    finally:
        torch.cuda.CUDAGraph = real
    path = os.path.join(tempfile.mkdtemp(), "step.dot")
    step.g1.debug_dump(path)
    txt = open(path).read()
    kinds = {}
    for m in re.finditer(r'label="([^"]*)"', txt):
        lab = m.group(1)
        k = "MEMSET" if "MEMSET" in lab.upper() else ("MEMCPY" if "MEMCPY" in lab.upper() else "KERNEL/OTHER")
        kinds[k] = kinds.get(k, 0) + 1
    memsets = sorted(set(re.findall(r'label="([^"]*(?:MEMSET|Memset|memset)[^"]*)"', txt)))[:5]
    return {"dot_bytes": len(txt), "node_labels_by_kind": kinds, "memset_label_samples": memsets}


def exp_autograd(accumulators_on_side, use_backward):
    """backward through a matmul chain inside a capture; leaves' grad accumulators created on a side stream or not"""
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(1)
    flat = torch.randn(4 * 256 * 256, generator=gen).to(dev)
    fg = torch.zeros_like(flat)
    ws = []
    for i in range(4):
        w = flat[i * 65536:(i + 1) * 65536].view(256, 256)
        w.requires_grad_(True)
        w.grad = fg[i * 65536:(i + 1) * 65536].view(256, 256)
        ws.append(w)
    x = torch.zeros(512, 256, device=dev)
    src = [torch.randn(512, 256, generator=gen).to(dev) for _ in range(3)]

    def step(inp):
        h = inp
        for w in ws:
            h = torch.tanh(h @ w)
        loss = (h * h).mean()
        fg.zero_()
        if use_backward:
            loss.backward()
        else:
            gs = torch.autograd.grad(loss, ws)
            for w, g_ in zip(ws, gs):
                w.grad.copy_(g_)
        return loss.detach()

    def build():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        if accumulators_on_side:
            with torch.cuda.stream(s):
                for _ in range(2):
                    step(x)
            torch.cuda.current_stream().wait_stream(s)
        else:
            for _ in range(2):
                step(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        out = {}
        with torch.cuda.graph(g):
            out["l"] = step(x)
        return g, lambda: [fg.clone(), out["l"].clone().view(1)]

    def eager(i):
        keep = fg.clone()
        l_ = step(src[i])
        r = [fg.clone(), l_.view(1)]
        fg.copy_(keep)
        return r

    return replays(build, lambda i: x.copy_(src[i]), eager)


def main():
    res = {"torch": torch.__version__, "hip": torch.version.hip}
    for op in ("sum", "norm", "colsum", "rowsum", "mean_dim"):
        for side in (False, True):
            res[f"reduction_{op}_warm_{'side' if side else 'current'}_stream"] = exp_reduction(side, op=op)
    res["raw_memset_in_capture"] = exp_memset()
    res["memset_sweep"] = exp_memset_sweep()
    res["forty_reductions_one_capture"] = exp_many_reductions()
    try:
        res["graphed_step_nodes"] = exp_graphed_step_nodes()
    except Exception as e:  # noqa: BLE001
        res["graphed_step_nodes"] = f"{type(e).__name__}: {e}"[:300]
    for side in (False, True):
        for bw in (True, False):
            k = f"autograd_{'backward' if bw else 'grad'}_accumulators_{'side' if side else 'current'}_stream"
            try:
                res[k] = exp_autograd(side, bw)
            except Exception as e:  # noqa: BLE001
                res[k] = f"{type(e).__name__}: {e}"[:200]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
