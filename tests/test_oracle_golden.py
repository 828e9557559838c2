"""Pin the CPU oracle to golden vectors captured from the reference itself
(tests/golden/make_golden.py) and to the mask tables in serenade/utils/masking.py:22-26,142-146.
CPU only."""
import json
import os

import numpy as np
import torch

from oracle import serenade_oracle as O
from tests._weights import hifigan_weights, serenade_weights, sub
from serenade_amd.utils.synth import synth_inputs

TOL = dict(rtol=2e-4, atol=2e-5)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, scale_tol=2e-5):
    """max |a-b| <= scale_tol * max|b|  (fp32 CPU summation-order noise only)."""
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= scale_tol * max(ref, 1e-6), (err, ref)


def test_state_dict_layout_matches_reference():
    from serenade_amd import _shapes as S
    from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))

    def tab(shapes):
        return [[k, list(s), str(dt).replace("torch.", "")] for k, (s, dt) in shapes.items()]

    assert sorted(tab(S.serenade_shapes(**SERENADE_PARAMS))) == sorted(ref["serenade"])
    assert sorted(tab(S.hifigan_shapes(**HIFIGAN_PARAMS, weight_norm=True))) == sorted(ref["hifigan_wn"])
    assert sorted(tab(S.hifigan_shapes(**HIFIGAN_PARAMS, weight_norm=False))) == sorted(ref["hifigan"])


def test_mask_known_answers(golden):
    g = golden("masks")
    # literal tables from the reference docstrings
    pad = [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]]
    assert O.make_pad_mask([5, 3, 2]).int().tolist() == pad
    assert O.make_non_pad_mask([5, 3, 2]).int().tolist() == [[1 - v for v in r] for r in pad]
    assert np.array_equal(O.make_pad_mask(T(g["lengths"])).numpy(), g["pad"])
    assert np.array_equal(O.make_non_pad_mask(T(g["lengths"])).numpy(), g["non_pad"])


def test_encoder(golden):
    g = golden("encoder")
    w = sub(serenade_weights(), "encoder.")
    close(O.conv1d_resnet(w, T(g["x"])), g["y"])


def test_gst(golden):
    g = golden("gst")
    w = sub(serenade_weights(), "gst.")
    close(O.reference_encoder(sub(w, "ref_enc."), T(g["speech"])), g["ref_embs"])
    close(O.style_encoder(w, T(g["speech"])), g["style"])


def _dec(golden, tag):
    g = golden("decoder_" + tag)
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    lens = g["lens"].tolist()
    mask = O.make_non_pad_mask(lens).unsqueeze(1)
    x, mu, spk, t = T(g["x"]), T(g["mu"]), T(g["spk"]), T(g["t"])
    temb = O.timestep_embedding(sub(w, "time_mlp."), O.sinusoidal_pos_emb(t, 242))
    close(temb, g["temb"])
    h = torch.cat([x, mu], dim=1)
    mf = mask.float()
    rb = sub(w, "down_blocks.0.0.")
    close(O.block1d(sub(rb, "block1."), h, mf), g["block1"])
    r1 = O.resnet_block1d(rb, h, mf, temb, spk)
    close(r1, g["resnet"])
    t1 = O.basic_transformer_block(sub(w, "down_blocks.0.1.0."), T(g["resnet"]).transpose(1, 2), mask[:, 0])
    close(t1, g["tfm"])
    close(O.decoder_forward(w, x, mask, mu, t, spk), g["out"], 5e-5)


def test_decoder_even_padded(golden):
    _dec(golden, "L48")


def test_decoder_odd_padded(golden):
    _dec(golden, "L65")


def test_euler_trace(golden):
    g = golden("euler_L48")
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1)
    trace = []
    out = O.solve_euler(w, T(g["z"]), T(g["mu"]), mask, T(g["spk"]), 10, trace=trace)
    for k in range(10):
        close(trace[k], g["trace"][k], 2e-4)
    close(out, g["out"], 2e-4)


def test_euler_20_steps_odd_padded(golden):
    """BASELINE configs[2] step count: 20 Euler steps, L = 57 (odd) with a padded row, vs the reference's solve_euler"""
    g = golden("euler20_L57")
    w = sub(serenade_weights(), "cfm_decoder.estimator.")
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1)
    trace = []
    out = O.solve_euler(w, T(g["z"]), T(g["mu"]), mask, T(g["spk"]), 20, trace=trace)
    close(trace[0], g["x1"], 2e-4)
    close(trace[9], g["x10"], 2e-4)
    close(out, g["out"], 2e-4)


def test_t_schedule_is_fp32_accumulated():
    ts, dts = O.t_schedule(10)
    t_span = torch.linspace(0, 1, 11)
    assert ts[0].item() == 0.0 and abs(ts[-1].item() - 0.9) < 1e-6
    assert dts[0].item() == (t_span[1] - t_span[0]).item()


def test_full_inference_chain(golden):
    g = golden("inference")
    w = serenade_weights()
    d = synth_inputs(1, 64, T_ref=16, seed=4321)
    z = (d["z"] / 0.667) * 0.667  # the reference multiplies randn by the temperature
    mel = O.serenade_inference(w, d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                               d["ref_logmel"], d["ref_midi"], d["ref_lft"], z)
    close(mel, g["mel_b1"], 2e-4)
    d = synth_inputs(2, 40, T_ref=16, seed=4322, lengths=[40, 29])
    z = (d["z"] / 0.667) * 0.667
    mel = O.serenade_inference(w, d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                               d["ref_logmel"], d["ref_midi"], d["ref_lft"], z)
    close(mel, g["mel_b2"], 2e-4)


def test_hifigan(golden):
    g = golden("hifigan")
    w, cfg = hifigan_weights()
    y = O.hifigan_forward(w, T(g["c"]), cfg)
    close(y, g["y"], 1e-4)
    gi = golden("inference")
    one = torch.ones(80)
    wave = O.vocoder_decode(w, T(gi["mel_b1"]), cfg, {"mean": 0 * one, "scale": one},
                            {"mean": 0 * one, "scale": one})
    assert wave.shape == (64 * 240,)
    assert (wave - T(g["wave_b1"])).abs().max().item() <= 1e-5


def test_hifigan_small_variant(golden):
    g = golden("hifigan_small")
    w, cfg = hifigan_weights(seed=1, small=True)
    close(O.hifigan_forward(w, T(g["c"]), cfg), g["y"], 1e-4)


def test_training_forward(golden):
    """a1' Serenade.forward on the reference's own random draws (captured by make_golden.py forward)."""
    g = golden("forward")
    w = serenade_weights()
    ret = O.serenade_forward(w, T(g["x"]), T(g["lens"]), T(g["logmel"]), T(g["midi"]), T(g["lft"]),
                             float(g["uniform"]), int(g["seg_start"]), T(g["t"]), T(g["z"]))
    close(ret["gauss_mel"], g["gauss_mel"])
    assert abs(ret["prior_loss"].item() - float(g["prior_loss"])) <= 1e-5 * abs(float(g["prior_loss"]))
    assert abs(ret["cfm_loss"].item() - float(g["cfm_loss"])) <= 1e-4 * abs(float(g["cfm_loss"]))
