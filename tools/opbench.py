#!/usr/bin/env python3
"""Per-op timing of the headline workload's plan (developer tool, GPU only): times every distinct conv_gemm call of
one Euler step + the vocoder and prints achieved TFLOP/s per shape, sorted by time share.

    python tools/opbench.py [tile-override]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from serenade_amd import ops  # noqa: E402


def sig(k):
    if "dilation" in k:  # fused HiFi-GAN residual unit: two convs of K = k * C
        return (k["n_batch"], k["T"], k["C"], 2 * k["k"] * k["C"], k["k"], False, False, k["dilation"], False, -2)
    return (k["n_batch"] * k.get("n_head", 1), k["T_out"], k["N"], len(k.get("taps", (0,))) * k["C_in"],
            len(k.get("taps", (0,))), bool(k.get("w_nmajor", False)), bool(k.get("geglu", False)),
            k.get("in_stride", 1), bool(k.get("gn_partials") is not None), k.get("pro_act", 0))


def time_ops(oplist, reps=20):
    agg = collections.OrderedDict()
    for op in oplist:
        if not isinstance(op, (ops.ConvOp, ops.ResUnitOp)):
            continue
        op()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            op()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        a = agg.setdefault(sig(op.kw), [0, 0.0])
        a[0] += 1
        a[1] += ms
    return agg


def report(title, agg):
    tot = sum(v[1] for v in agg.values())
    print(f"== {title}: {tot:.2f} ms total")
    print(f"{'Z':>4} {'T_out':>7} {'N':>5} {'K':>6} taps nmaj geglu str gn act | cnt  ms_each   TF/s  share")
    for k, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        Z, T, N, K, taps, nmaj, geglu, st, gn, act = k
        fl = 2.0 * Z * T * N * K
        print(f"{Z:4d} {T:7d} {N:5d} {K:6d} {taps:4d} {int(nmaj):4d} {int(geglu):5d} {st:3d} {int(gn):2d} {act:3d} | "
              f"{cnt:3d} {ms / cnt:8.3f} {fl * cnt / (ms * 1e-3) / 1e12:6.1f} {ms / tot * 100:6.1f}%")


def sweep(oplist, reps=5):
    """time every distinct shape under every admissible tile id"""
    seen = {}
    for op in oplist:
        if isinstance(op, ops.ConvOp) and sig(op.kw) not in seen:
            seen[sig(op.kw)] = op
    print(f"{'Z':>4} {'T_out':>7} {'N':>5} {'K':>6} taps nmaj geglu |  auto    t1     t4     t6     t7     t8     t9    t10    t11   old7   old9  (TF/s)")
    for k, op in seen.items():
        Z, T, N, K, taps, nmaj, geglu, st, gn, act = k
        fl = 2.0 * Z * T * N * K
        res = []
        for tile in (0, 1, 4, 6, 7, 8, 9, 10, 11, -7, -9):
            if tile in (2, 5) and nmaj:
                res.append(float("nan"))
                continue
            if geglu and abs(tile) in (4, 5, 7, 10, 11):
                res.append(float("nan"))
                continue
            o2 = ops.ConvOp(**dict(op.kw, tile=abs(tile), no_halo=5 if tile < 0 else op.kw.get("no_halo", False)))
            try:
                o2()
            except RuntimeError:  # a tile id the kernel that takes this shape does not have
                res.append(float("nan"))
                continue
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                o2()
            e.record()
            torch.cuda.synchronize()
            res.append(fl / (s.elapsed_time(e) / reps * 1e-3) / 1e12)
        print(f"{Z:4d} {T:7d} {N:5d} {K:6d} {taps:4d} {int(nmaj):4d} {int(geglu):5d} | " +
              " ".join(f"{r:6.1f}" for r in res))


def main():
    if "--no-halo" in sys.argv:
        ops.NO_HALO = True
    import serenade_amd
    serenade_amd.set_precision("fp32" if "--fp32" in sys.argv else ("bf16x6" if "--bf16x6" in sys.argv else "bf16x3"))
    dev = torch.device("cuda:0")
    model, voc, sd, gsd = bench.build_models(dev)
    B, T, Tr = (int(os.environ.get(k, d)) for k, d in (("SRN_B", bench.B_PER_GPU), ("SRN_T", bench.T_SRC), ("SRN_TREF", bench.T_REF)))
    from serenade_amd.utils.synth import synth_inputs
    d = synth_inputs(B, T, T_ref=Tr, seed=1235)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}
    mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                          g["ref_midi"], g["ref_lft"], noise=g["z"])
    voc.decode_batch(mel if mel.dim() == 3 else mel.unsqueeze(0))
    torch.cuda.synchronize()
    pl = model.cfm_decoder.estimator.plan(B, T + Tr, 10, euler=True)
    if "--sweep" in sys.argv:
        sweep(pl.steps[0])
        sweep(voc.model.plan(B, T).ops)
        return
    report("one Euler step (x10 per utterance batch)", time_ops(pl.steps[0]))
    vp = voc.model.plan(B, T)
    report("HiFi-GAN forward", time_ops(vp.ops))


if __name__ == "__main__":
    main()
