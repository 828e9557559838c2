#!/usr/bin/env python3
"""HBM-side traffic of the contraction kernels over the headline workload (developer tool).

  step 1 (GPU box, one counter per pass as MI355X_MICROARCH.md prescribes):
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 tools/plainloop.py 1
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 tools/plainloop.py 1
  step 2:
      python tools/traffic.py gpurun_out/pmc_f gpurun_out/pmc_w [--algorithmic]

Prints, for the srn_conv_gemm kernels (conv_fast / conv_halo / conv_strip / conv_gemm), launches, fetched and written
bytes per launch.  Units and corrections: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B for wide coalesced reads, so it is doubled (the guide's HBM section); WRITE_SIZE is
exact for 16-B-per-lane stores, uncalibrated for this kernel's 4-B-per-lane stores (reported as is).
--algorithmic additionally builds the plans (needs the GPU) and sums the operand bytes every launch must move once
(A + weights + output + residuals) for comparison.
"""
import csv
import glob
import sys


def conv_counter(path, counter):
    tot, n = 0.0, 0
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "conv_" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"])
                n += 1
    return tot, n


def algorithmic_bytes():
    import os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from serenade_amd import ops
    from serenade_amd.utils.synth import synth_inputs
    dev = torch.device("cuda:0")
    model, voc, sd, gsd = bench.build_models(dev)
    d = synth_inputs(bench.B_PER_GPU, bench.T_SRC, T_ref=bench.T_REF, seed=1235)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}
    seen = []
    orig = ops.ConvOp.__call__

    def spy(self, stream=None):
        seen.append(self.kw)
        return orig(self, stream)

    ops.ConvOp.__call__ = spy
    mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"],
                          g["ref_midi"], g["ref_lft"], n_timesteps=bench.N_EULER, noise=g["z"])
    voc.decode_batch(mel)
    torch.cuda.synchronize()
    ops.ConvOp.__call__ = orig
    rd = wr = 0
    for k in seen:
        Z = k["n_batch"] * k.get("n_head", 1)
        taps = len(k.get("taps", (0,)))
        n_out = k.get("N_out", 0) or k["N"]
        rd += Z * k["T_in"] * k["C_in"] * 4  # activations, read once
        per_z = bool(k.get("w_bs", 0) or k.get("w_hs", 0))
        rd += (Z if per_z else 1) * k["N"] * taps * k["C_in"] * 4  # weights / B operand
        out_b = Z * k["T_out"] * n_out * 4
        wr += out_b
        rd += out_b * ((k.get("res") is not None) + (k.get("res2") is not None))
    return len(seen), rd, wr


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    f_kib, nf = conv_counter(args[0], "FETCH_SIZE")
    w_kib, nw = conv_counter(args[1], "WRITE_SIZE")
    fetched = 2.0 * f_kib * 1024.0  # gfx950: 128-B requests tallied at 64 B
    written = w_kib * 1024.0
    print(f"contraction launches: {nf} (fetch pass) / {nw} (write pass)")
    print(f"FETCH_SIZE x2: {fetched / 1e9:.2f} GB total, {fetched / max(nf, 1) / 1e6:.2f} MB per launch")
    print(f"WRITE_SIZE   : {written / 1e9:.2f} GB total, {written / max(nw, 1) / 1e6:.2f} MB per launch")
    print(f"traffic per launch: {(fetched / max(nf, 1) + written / max(nw, 1)) / 1e6:.2f} MB")
    if "--algorithmic" in sys.argv:
        n, rd, wr = algorithmic_bytes()
        print(f"algorithmic: {n} launches per step, read {rd / 1e9:.2f} GB, write {wr / 1e9:.2f} GB per step, "
              f"{(rd + wr) / n / 1e6:.2f} MB per launch")


if __name__ == "__main__":
    main()
