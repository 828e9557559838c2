"""Host-logic tests (CPU): weight packing, buffer plans, strides, tap tables and phase decomposition of the
product modules, executed through the test-only C-ABI emulator (tests/_emulator.py) and compared with the
oracle / the reference's golden vectors.  The HIP kernels themselves are tested by the -m gpu suite."""
import numpy as np
import pytest
import torch

from oracle import serenade_oracle as O
from serenade_amd import models, vocoder
from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict, synth_inputs
from tests import _emulator
from tests._weights import hifigan_weights, serenade_weights, sub


def T(a):
    return torch.from_numpy(np.asarray(a))


def nerr(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.fixture(scope="module")
def model():
    m = models.Serenade(**SERENADE_PARAMS)
    m.load_state_dict(serenade_weights())
    return m.eval()


def test_state_dict_keys_match_tables(model):
    assert list(model.state_dict().keys()) == list(serenade_weights().keys())


@pytest.mark.parametrize("tag", ["L48", "L65"])
def test_decoder_forward(model, golden, tag):
    g = golden("decoder_" + tag)
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1)
    with _emulator.installed():
        out = model.cfm_decoder.estimator(T(g["x"]), mask, T(g["mu"]), T(g["t"]), T(g["spk"]))
    assert nerr(out, g["out"]) < 5e-5


@pytest.mark.parametrize("pairs", [1, 3, 4])
def test_attention_chunking_is_invisible(model, golden, monkeypatch, pairs):
    """the score buffer holds `pairs` (batch, head) pairs: head-runs inside an item (1, 3) and item-runs (4 = one
    item's 4 heads) must give the decoder output of the unchunked plan (padded batch of 2, odd L)"""
    g = golden("decoder_L65")
    L = 65
    monkeypatch.setattr(models, "S_BUDGET", pairs * L * 96 * 4)  # rup(65, 32) = 96
    est = model.cfm_decoder.estimator
    est._plans.clear()
    assert len(models.attention_chunks(2, 4, L * 96 * 4, pairs * L * 96 * 4)) == {1: 8, 3: 4, 4: 2}[pairs]
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1)
    with _emulator.installed():
        out = est(T(g["x"]), mask, T(g["mu"]), T(g["t"]), T(g["spk"]))
    est._plans.clear()
    assert nerr(out, g["out"]) < 5e-5


def test_solve_euler(model, golden):
    g = golden("euler_L48")
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1)
    with _emulator.installed():
        out = model.cfm_decoder.solve_euler(T(g["z"]), torch.linspace(0, 1, 11), T(g["mu"]), mask, T(g["spk"]))
    assert nerr(out, g["out"]) < 2e-4


def test_encoder_and_gst(model, golden):
    g = golden("encoder")
    with _emulator.installed():
        y = model.encoder(T(g["x"]), None)
    assert nerr(y, g["y"]) < 2e-5
    g = golden("gst")
    with _emulator.installed():
        s = model.gst(T(g["speech"]))
    assert nerr(s, g["style"]) < 2e-5


def test_inference_chain(model, golden):
    g = golden("inference")
    d = synth_inputs(1, 64, T_ref=16, seed=4321)
    z = (d["z"] / 0.667) * 0.667
    with _emulator.installed():
        mel = model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                              d["ref_logmel"], d["ref_midi"], d["ref_lft"], noise=z)
    assert mel.shape == (64, 80)
    assert nerr(mel, g["mel_b1"]) < 2e-4
    d = synth_inputs(2, 40, T_ref=16, seed=4322, lengths=[40, 29])
    z = (d["z"] / 0.667) * 0.667
    with _emulator.installed():
        mel = model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                              d["ref_logmel"], d["ref_midi"], d["ref_lft"], noise=z)
    assert nerr(mel, g["mel_b2"]) < 2e-4


def test_inference_uses_cpu_generator_like_reference(model):
    d = synth_inputs(1, 24, T_ref=16, seed=7)
    with _emulator.installed():
        torch.manual_seed(11)
        a = model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                            d["ref_logmel"], d["ref_midi"], d["ref_lft"])
        torch.manual_seed(11)
        z = torch.randn((1, 80, 40)) * 0.667
        b = model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                            d["ref_logmel"], d["ref_midi"], d["ref_lft"], noise=z)
    assert torch.equal(a, b)


def test_exact_ragged_batch_equals_b1_runs(model):
    """inference_ragged: three items with different source and prompt lengths in one padded batch; every item equals
    its own B = 1 inference() and the oracle (per-item GroupNorm statistics, reflection at the item's end, offsets)"""
    shapes = [(24, 16), (17, 21), (31, 9)]
    ds = [synth_inputs(1, t, T_ref=r, seed=70 + i) for i, (t, r) in enumerate(shapes)]
    items = [tuple(d[k][0] for k in ("x", "midi", "lft", "ref_x", "ref_logmel", "ref_midi", "ref_lft")) for d in ds]
    with _emulator.installed():
        outs = model.inference_ragged(items, noises=[d["z"][0] for d in ds])
        singles = [model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                                   d["ref_logmel"], d["ref_midi"], d["ref_lft"], noise=d["z"]) for d in ds]
        with pytest.raises(ValueError):
            model.inference_ragged([])
    w = serenade_weights()
    for (t, _), o, s, d in zip(shapes, outs, singles, ds):
        assert o.shape == (t, 80) and nerr(o, s) < 2e-5
        ref = O.serenade_inference(w, d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                                   d["ref_logmel"], d["ref_midi"], d["ref_lft"], d["z"], n_timesteps=10)
        assert nerr(o, ref) < 1e-4


def _gen(small=False, wn=True):
    w, params = hifigan_weights(seed=1 if small else 0, small=small)
    g = vocoder.HiFiGANGenerator(**params)
    from serenade_amd import _shapes
    sd = fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**params, weight_norm=True)), seed=1 if small else 0)
    g.load_state_dict(sd)
    return g.eval(), w, params


def test_hifigan_forward_and_weight_norm_removal(golden):
    g = golden("hifigan")
    gen, w, params = _gen()
    with _emulator.installed():
        y = gen(T(g["c"]))  # weight norm folded on the fly
        gen.remove_weight_norm()
        assert sorted(gen.state_dict().keys()) == sorted(w.keys())
        y2 = gen(T(g["c"]))
    assert y.shape == (2, 1, 2880)
    assert nerr(y, g["y"]) < 1e-4 and nerr(y2, g["y"]) < 1e-4


def test_hifigan_small_variant(golden):
    g = golden("hifigan_small")
    gen, w, params = _gen(small=True)
    with _emulator.installed():
        y = gen(T(g["c"]))
    assert nerr(y, g["y"]) < 1e-4


def test_vocoder_decode(golden):
    gi, gh = golden("inference"), golden("hifigan")
    gen, w, params = _gen()
    one = np.ones(80, dtype=np.float32)
    rng = np.random.default_rng(3)
    stats = {"mean": rng.standard_normal(80).astype(np.float32) * 0.1, "scale": one + 0.1}
    trg = {"mean": rng.standard_normal(80).astype(np.float32) * 0.1, "scale": one * 0.9}
    with _emulator.installed():
        voc = vocoder.Vocoder.from_generator(gen, {"sampling_rate": 24000}, {"mean": 0 * one, "scale": one},
                                             torch.device("cpu"), trg_stats={"mean": 0 * one, "scale": one})
        wave, sr = voc.decode(T(gi["mel_b1"]))
        assert sr == 24000 and wave.shape == (64 * 240,)
        assert (wave - T(gh["wave_b1"])).abs().max().item() < 2e-5
        voc2 = vocoder.Vocoder.from_generator(gen, {"sampling_rate": 24000}, stats, torch.device("cpu"), trg_stats=trg)
        wb = voc2.decode_batch(T(gi["mel_b2"]))
    ref = O.vocoder_decode(w, T(gi["mel_b2"]), params, {k: T(v) for k, v in stats.items()},
                           {k: T(v) for k, v in trg.items()})
    assert wb.shape == ref.shape == (2, 40 * 240)
    assert (wb - ref).abs().max().item() < 2e-5


def test_cpu_tensors_are_refused_without_emulator(model):
    d = synth_inputs(1, 8, T_ref=8, seed=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.inference(d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"], d["ref_logmel"],
                        d["ref_midi"], d["ref_lft"])


def test_training_forward_values(model, golden):
    """a1' Serenade.forward: loss values on the reference's own random draws (per-sample time embedding)."""
    g = golden("forward")
    draws = {"uniform": float(g["uniform"]), "seg_start": int(g["seg_start"]), "t": T(g["t"]), "z": T(g["z"])}
    with _emulator.installed():
        ret = model(T(g["x"]), T(g["lens"]), T(g["logmel"]), T(g["midi"]), T(g["lft"]), draws=draws)
    assert set(ret) == {"gauss_mel", "prior_loss", "cfm_loss"}
    assert nerr(ret["gauss_mel"], g["gauss_mel"]) < 2e-5
    assert abs(ret["prior_loss"].item() / float(g["prior_loss"]) - 1) < 1e-5
    assert abs(ret["cfm_loss"].item() / float(g["cfm_loss"]) - 1) < 1e-4
