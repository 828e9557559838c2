"""Golden vectors for the in-tree arithmetic of the reference's post-processing stage and more mask
KATs -- run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_post.py

`serenade/bin/ssc_postprocessing.py` imports hydra / omegaconf / librosa / pysptk / pyworld / soundfile / h5py / sifigan
at module level; none of them is in the image.  The functions captured here (`convert_continuos_f0` :51-72 and the
`np.interp` length match :159-164, re-run below on the same inputs) use only numpy, copy and scipy's `interp1d`, so the
absent packages are replaced by EMPTY modules (plus a pass-through `hydra.main` decorator): nothing they would compute
is exercised or captured.  WORLD / SPTK / SiFiGAN arithmetic stays parity-unpinned.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_harness  # noqa: E402

_ref_harness.install()


def _empty(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _passthrough_main(*a, **k):
    return lambda f: f


_empty("hydra", main=_passthrough_main)
_empty("hydra.utils", instantiate=None, to_absolute_path=None)
_empty("omegaconf", DictConfig=dict)
for _n in ("librosa", "pysptk", "pyworld", "tqdm"):
    _empty(_n, tqdm=None)
_empty("sifigan")
_empty("sifigan.utils")
_empty("sifigan.utils.features", SignalGenerator=None, dilated_factor=None)

from serenade.bin.ssc_postprocessing import ALPHA, convert_continuos_f0  # noqa: E402
from serenade.utils.masking import make_non_pad_mask, make_pad_mask  # noqa: E402

import torch  # noqa: E402


def contour(rng, n, voiced_runs):
    f0 = np.zeros(n)
    for lo, hi in voiced_runs:
        f0[lo:hi] = 220.0 * 2.0 ** (rng.uniform(-1, 1) + 0.1 * np.cumsum(rng.standard_normal(hi - lo)) / 12.0)
    return f0


def main():
    rng = np.random.default_rng(2025)
    out = {"alpha_24000": np.float64(ALPHA[24000])}
    cases = {
        "mixed": contour(rng, 97, [(5, 30), (41, 42), (60, 90)]),
        "voiced_edges": contour(rng, 40, [(0, 12), (25, 40)]),
        "all_voiced": contour(rng, 33, [(0, 33)]),
        "single": contour(rng, 20, [(7, 8)]),
        "all_zero": np.zeros(16),
    }
    # a repeated start / end value inside the contour: the reference pads up to the FIRST index holding the start
    # value and from the LAST index holding the end value (np.where(...)[0][0] / [0][-1], :61-64)
    rep = contour(rng, 50, [(4, 20), (30, 44)])
    rep[35] = rep[4]
    rep[10] = rep[43]
    cases["repeated_values"] = rep
    for name, f0 in cases.items():
        uv, cf0, flag = convert_continuos_f0(f0.copy())
        out[f"{name}_f0"] = f0
        out[f"{name}_uv"] = np.asarray(uv)
        out[f"{name}_cf0"] = np.asarray(cf0, dtype=np.float64)
        out[f"{name}_flag"] = np.bool_(flag)
    # length match (:159-164): linspace resampling + clamp at 0, as the script does it
    for name, (n_in, n_out) in {"up": (51, 103), "down": (80, 37), "same": (19, 19)}.items():
        f0_ = contour(rng, n_in, [(3, n_in // 2), (n_in // 2 + 4, n_in - 2)]).reshape(-1, 1)
        x_orig = np.arange(len(f0_))
        x_new = np.linspace(0, len(f0_) - 1, n_out)
        y = np.maximum(np.interp(x_new, x_orig, f0_.ravel()), 0)
        out[f"resample_{name}_in"] = f0_
        out[f"resample_{name}_out"] = y
    np.savez_compressed(os.path.join(HERE, "postproc_f0.npz"), **out)
    print("wrote postproc_f0", len(out), "arrays")

    # ---- more mask KATs: the xs / length_dim / maxlen forms (serenade/utils/masking.py:28-88)
    lengths = [5, 3, 2]
    m = {"lengths": np.array(lengths)}
    m["xs_3_4_6_dim-1"] = make_pad_mask(lengths, torch.zeros(3, 4, 6)).numpy()
    m["xs_3_6_6_dim1"] = make_pad_mask(lengths, torch.zeros(3, 6, 6), 1).numpy()
    m["xs_3_6_6_dim2"] = make_pad_mask(lengths, torch.zeros(3, 6, 6), 2).numpy()
    m["xs_3_2_5_4_dim-2"] = make_pad_mask(lengths, torch.zeros(3, 2, 5, 4), -2).numpy()
    m["maxlen7"] = make_pad_mask(lengths, maxlen=7).numpy()
    m["tensor_lengths"] = make_pad_mask(torch.tensor(lengths)).numpy()
    m["non_pad_xs_3_4_6"] = make_non_pad_mask(lengths, torch.zeros(3, 4, 6)).numpy()
    m["non_pad_xs_3_6_6_dim1"] = make_non_pad_mask(lengths, torch.zeros(3, 6, 6), 1).numpy()
    np.savez_compressed(os.path.join(HERE, "masks_xs.npz"), **m)
    print("wrote masks_xs", {k: v.shape for k, v in m.items()})


if __name__ == "__main__":
    main()
