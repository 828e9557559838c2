"""Row a9 — SiFiGAN V2 generator (parity UNPINNED: the upstream `sifigan` package is not in the reference tree).
The HIP implementation is checked against this repo's own restatement of the published algorithm
(oracle/sifigan_oracle.py): on CPU through the C-ABI emulator (host logic) and, marked gpu, on the MI355X."""
import numpy as np
import pytest
import torch

import serenade_amd
from oracle import sifigan_oracle as SO
from serenade_amd import _shapes, sifigan
from serenade_amd.utils.synth import fill_state_dict
from tests import _emulator
from tests._weights import fold_weight_norm

SMALL = dict(sifigan.DEFAULT_PARAMS, channels=64, upsample_scales=(3, 2), upsample_kernel_sizes=(6, 4),
             source_network_params=dict(resblock_kernel_size=3, resblock_dilations=[(1,), (1, 2)],
                                        use_additional_convs=True),
             filter_network_params=dict(resblock_kernel_sizes=(3, 5), resblock_dilations=[(1, 3), (1, 3)],
                                        use_additional_convs=False))


def _inputs(cfg, B, T, seed=0):
    rng = np.random.default_rng(seed)
    hop = int(np.prod(cfg["upsample_scales"]))
    c = torch.from_numpy(rng.standard_normal((B, cfg["in_channels"], T)).astype(np.float32))
    f0 = rng.uniform(100, 400, (B, 1, T))
    x = torch.from_numpy((0.1 * np.sin(np.cumsum(np.repeat(2 * np.pi * f0 / 24000, hop, axis=2), axis=2)))
                         .astype(np.float32))
    d = []
    for df, us in zip((0.5, 1, 4, 8), np.cumprod(cfg["upsample_scales"])):
        # dilated_factor = sample_rate / f0 / dense_factor (ssc_postprocessing.py:201-210 builds these with np.repeat)
        d.append(torch.from_numpy(np.repeat(24000.0 / f0 / df / 16.0, us, axis=2).astype(np.float32)))
    return x, c, d


def _model(cfg, seed=0):
    g = sifigan.SiFiGANGenerator(**cfg)
    sd = fill_state_dict(_shapes.as_meta(sifigan.sifigan_shapes(**cfg)), seed=seed)
    g.load_state_dict(sd)
    return g.eval(), fold_weight_norm(sd)


def nerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def test_sifigan_plan_matches_restatement_cpu():
    g, w = _model(SMALL)
    x, c, d = _inputs(SMALL, 2, 11)
    y_ref, e_ref = SO.sifigan_forward(w, x, c, d, SMALL)
    with _emulator.installed():
        y, e = g(x, c, d)
        g.remove_weight_norm()
        y2, _ = g(x, c, d)
    assert y.shape == (2, 1, 66) and e.shape == (2, 1, 66)
    assert nerr(y, y_ref) < 1e-4 and nerr(e, e_ref) < 1e-4 and nerr(y2, y_ref) < 1e-4
    assert sorted(g.state_dict().keys()) == sorted(w.keys())


def test_pd_indexing_known_answer():
    x = torch.arange(8.0).view(1, 1, 8)
    d = torch.tensor([0.0, 1.0, 1.4, 1.5, 2.5, 3.0, 0.5, 9.0]).view(1, 1, 8)
    xp, xf = SO.pd_indexing(x, d, 1)
    # r = round-half-even(d) = 0 1 1 2 2 3 0 9
    assert xp.view(-1).tolist() == [0.0, 0.0, 1.0, 1.0, 2.0, 2.0, 6.0, 0.0]
    assert xf.view(-1).tolist() == [0.0, 2.0, 3.0, 5.0, 6.0, 0.0, 6.0, 0.0]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_sifigan_gpu_default_config(precision):
    """full-width config of serenade/bin/sifigan_config/generator/sifigan.yaml on the MI355X vs the restatement"""
    serenade_amd.set_precision(precision)
    dev = torch.device("cuda:0")
    cfg = sifigan.DEFAULT_PARAMS
    g, w = _model(cfg, seed=2)
    x, c, d = _inputs(cfg, 2, 24, seed=3)
    y_ref, e_ref = SO.sifigan_forward(w, x, c, d, cfg)
    y, e = g.to(dev)(x.to(dev), c.to(dev), [t.to(dev) for t in d])
    tol = 1e-4 if precision == "fp32" else 1e-3
    assert y.shape == (2, 1, 24 * 120)
    assert nerr(y, y_ref) < tol and nerr(e, e_ref) < tol
    serenade_amd.set_precision("fp32")  # the package default
