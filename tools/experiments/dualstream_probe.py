#!/usr/bin/env python3
"""Probe (GPU): the headline batch as TWO half batches on two HIP streams (two model instances, so two sets of plan buffers)
against one B = 8 batch: do the two streams' kernels fill each other's launch gaps and fill / drain phases?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import serenade_amd
from serenade_amd.utils.synth import synth_inputs

serenade_amd.set_precision("fp32")
dev = torch.device("cuda:0")
m1, v1, _, _ = bench.build_models(dev)
m2, v2, _, _ = bench.build_models(dev)


def mk(model, voc, B, seed):
    d = synth_inputs(B, 1024, T_ref=256, seed=seed)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}
    def step():
        mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                              g["ref_midi"], g["ref_lft"], n_timesteps=10, noise=g["z"])
        return voc.decode_batch(mel)
    return step


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = mk(m1, v1, 8, 1235)
print(f"one B=8 batch on one stream: {timeit(one):.1f} ms")
for Bh in (4, 2):
    parts = [mk(m, v, Bh, 1235 + i) for i, (m, v) in enumerate(((m1, v1), (m2, v2)))]
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    def two():
        for _ in range(8 // (2 * Bh)):
            for st, p in zip(s, parts):
                with torch.cuda.stream(st):
                    p()
    def serial():
        for _ in range(8 // (2 * Bh)):
            for p in parts:
                p()
    print(f"8 utterances as B={Bh} batches: serial on one stream {timeit(serial):.1f} ms, pairs on two streams {timeit(two):.1f} ms")
