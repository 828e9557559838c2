"""BASELINE.json's configurations on the GPU, through the product path (Serenade.inference + Vocoder.decode_batch over
libserenade_hip.so), against the CPU oracle on identical inputs and explicit noise, in both contraction modes:

    C1  B=1, T=256, T_ref=256 (L=512), 10 Euler steps + HiFi-GAN          -- the reference's own CLI shape
    C3  B=8, T=1024, T_ref=256 (L=1280), 20 Euler steps + HiFi-GAN        -- configs[2] step count
    C5  B=4 (one GPU's share of 32), T=4096 (L=4352), 10 steps + HiFi-GAN -- long form
    (C2 = C3's shape at 10 steps is tests/test_hip_parity.py::test_full_size_batch_parity_and_properties;
     C4 = 8-way sharding + RCCL gather needs 8 GPUs: its single-GPU rehearsal is test_rccl_group_of_one below and the
     gloo world-size-2 test in tests/test_parallel_gloo.py)

Gates (north star): mel <= 1e-3 relative to the oracle's max, waveform <= 1e-4 absolute.  The oracle results are
computed once and shared by the two precision arms.  Also here: reloading weights into a model that has already run
(stale-plane regression) and the decode CLI on the GPU lives in tests/test_cli.py.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import serenade_amd
from oracle import serenade_oracle as O
from serenade_amd import _lib, _shapes, models, vocoder
from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict, synth_inputs
from tests._weights import hifigan_weights, serenade_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEL_RTOL, WAVE_ATOL = 1e-3, 1e-4
_ORACLE = {}


@pytest.fixture(autouse=True, params=["fp32", "bf16x3", "bf16x6"])
def precision(request):
    serenade_amd.set_precision(request.param)
    yield request.param
    serenade_amd.set_precision("fp32")  # the package default


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need an MI355X"
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    m = models.Serenade(**SERENADE_PARAMS)
    m.load_state_dict(serenade_weights())
    return m.eval().to(dev)


@pytest.fixture(scope="module")
def voc(dev):
    g = vocoder.HiFiGANGenerator(**HIFIGAN_PARAMS)
    g.load_state_dict(fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**HIFIGAN_PARAMS, weight_norm=True)),
                                      seed=0))
    one = np.ones(80, dtype=np.float32)
    ident = {"mean": 0 * one, "scale": one}
    return vocoder.Vocoder.from_generator(g, {"sampling_rate": 24000}, ident, dev, trg_stats=ident)


def nerr(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def infer(model, d, dev, **kw):
    g = lambda k: d[k].to(dev)
    return model.inference(g("x"), d["lengths"], g("midi"), g("lft"), g("ref_x"), d["ref_lengths"], g("ref_logmel"),
                           g("ref_midi"), g("ref_lft"), **kw)


def oracle_chain(key, one, n):
    """(mel, wave) of one utterance through the CPU oracle; cached so the second precision arm reuses it"""
    if key not in _ORACLE:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)))))
        mel = O.serenade_inference(serenade_weights(), one["x"], one["lengths"], one["midi"], one["lft"],
                                   one["ref_x"], one["ref_lengths"], one["ref_logmel"], one["ref_midi"],
                                   one["ref_lft"], one["z"], n_timesteps=n)
        id80 = {"mean": torch.zeros(80), "scale": torch.ones(80)}
        gw, gp = hifigan_weights()
        _ORACLE[key] = (mel, O.vocoder_decode(gw, mel, gp, id80, id80))
    return _ORACLE[key]


def test_c1_single_clip_exact_shape(dev, model, voc):
    """configs[0]: one 2.6 s clip, prompt as long as the source (B=1, T=256, T_ref=256, n=10) + vocoder"""
    d = synth_inputs(1, 256, T_ref=256, seed=1234)
    mel = infer(model, d, dev, noise=d["z"])
    assert mel.shape == (256, 80)
    wave, sr = voc.decode(mel)
    assert sr == 24000 and wave.shape == (256 * 240,)
    ref_mel, ref_wave = oracle_chain("c1", d, 10)
    assert nerr(mel, ref_mel) < MEL_RTOL
    assert (wave.cpu() - ref_wave).abs().max().item() < WAVE_ATOL


def test_c3_batch8_20_euler_steps(dev, model, voc, golden):
    """configs[2]: B=8, T=1024, 20 Euler steps + vocoder; utterance 5 vs the oracle, and the reference's own 20-step
    solve_euler fixture (odd, padded length)"""
    g = golden("euler20_L57")
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    mask = O.make_non_pad_mask(g["lens"].tolist()).unsqueeze(1).to(dev)
    out = model.cfm_decoder.solve_euler(t(g["z"]), torch.linspace(0, 1, 21), t(g["mu"]), mask, t(g["spk"]))
    assert nerr(out, g["out"]) < MEL_RTOL
    B, Tn = 8, 1024
    d = synth_inputs(B, Tn, T_ref=256, seed=1236)
    mel = infer(model, d, dev, n_timesteps=20, noise=d["z"])
    assert mel.shape == (B, Tn, 80) and torch.isfinite(mel).all()
    wave = voc.decode_batch(mel)
    i = 5
    one = {k: v[i:i + 1] for k, v in d.items()}
    ref_mel, ref_wave = oracle_chain("c3", one, 20)
    assert nerr(mel[i], ref_mel) < MEL_RTOL
    assert (wave[i].cpu() - ref_wave).abs().max().item() < WAVE_ATOL


def test_c5_long_form_share_of_one_gpu(dev, model, voc):
    """configs[4] per GPU: B=4, T=4096 (L=4352), 10 steps + vocoder: finite, bit-deterministic, batch-independent,
    and utterance 0 equals the CPU oracle's full chain"""
    B, Tn = 4, 4096
    d = synth_inputs(B, Tn, T_ref=256, seed=1238)
    mel = infer(model, d, dev, noise=d["z"])
    assert mel.shape == (B, Tn, 80) and torch.isfinite(mel).all()
    assert torch.equal(mel, infer(model, d, dev, noise=d["z"]))
    wave = voc.decode_batch(mel)
    assert wave.shape == (B, Tn * 240) and torch.isfinite(wave).all()
    one = {k: v[2:3] for k, v in d.items()}
    assert nerr(infer(model, one, dev, noise=one["z"]), mel[2]) < 1e-5
    one = {k: v[0:1] for k, v in d.items()}
    ref_mel, ref_wave = oracle_chain("c5", one, 10)
    assert nerr(mel[0], ref_mel) < MEL_RTOL
    assert (wave[0].cpu() - ref_wave).abs().max().item() < WAVE_ATOL


def test_c5_mfma_attention_option(dev, model, voc, precision):
    """configs[4]'s "MFMA attention": Q K^T and P V on the bf16 matrix cores (split-bf16, fp32 accumulate), every conv /
    linear in exact fp32 -- holds the north star's gates at T = 4096 and differs from the all-fp32 run"""
    if precision != "fp32":
        pytest.skip("one arm: main precision fp32, attention precision switched")
    B, Tn = 1, 4096
    d = synth_inputs(4, Tn, T_ref=256, seed=1238)
    one = {k: v[0:1] for k, v in d.items()}
    ref_mel, ref_wave = oracle_chain("c5", one, 10)
    plain = infer(model, one, dev, noise=one["z"])
    for attn in ("bf16x3", "bf16x6"):
        serenade_amd.set_attention_precision(attn)
        try:
            assert serenade_amd.get_attention_precision() == attn
            mel = infer(model, one, dev, noise=one["z"])
            wave = voc.decode_batch(mel.unsqueeze(0) if mel.dim() == 2 else mel)
        finally:
            serenade_amd.set_attention_precision(None)
        assert mel.shape == plain.shape and not torch.equal(mel, plain)
        assert nerr(mel.reshape(ref_mel.shape), ref_mel) < MEL_RTOL, attn
        assert (wave.reshape(-1).cpu() - ref_wave.reshape(-1)).abs().max().item() < WAVE_ATOL, attn


def test_reloading_weights_into_a_model_that_has_run(dev):
    """split-bf16 weight planes are cached per weight tensor; load_state_dict() copies into the live
    parameters (same address), so a stale cache would contract the new checkpoint with the old planes.  Run with
    seed-0 weights, reload seed-1 weights in place, run again, compare with a fresh seed-1 model (bit-exact)."""
    d = synth_inputs(1, 48, T_ref=16, seed=31)
    shapes = _shapes.as_meta(_shapes.serenade_shapes(**SERENADE_PARAMS))
    m = models.Serenade(**SERENADE_PARAMS)
    m.load_state_dict(fill_state_dict(shapes, seed=0))
    m = m.eval().to(dev)
    a0 = infer(m, d, dev, noise=d["z"])
    m.load_state_dict(fill_state_dict(shapes, seed=1))
    a1 = infer(m, d, dev, noise=d["z"])
    fresh = models.Serenade(**SERENADE_PARAMS)
    fresh.load_state_dict(fill_state_dict(shapes, seed=1))
    b1 = infer(fresh.eval().to(dev), d, dev, noise=d["z"])
    assert not torch.equal(a0, a1)
    assert torch.equal(a1, b1)


def test_exact_ragged_batch_equals_b1_runs(dev, model):
    """Serenade.inference_ragged: items with different source AND prompt lengths in one padded batch, each equal to its
    own B = 1 `inference` call (GroupNorm over valid rows, reflection at the item's end, per-item row offsets)"""
    shapes = [(200, 96), (131, 160), (257, 33), (64, 64)]
    items, noises, singles = [], [], []
    for i, (t, r) in enumerate(shapes):
        d = synth_inputs(1, t, T_ref=r, seed=500 + i)
        g = lambda k: d[k][0].to(dev)
        items.append((g("x"), g("midi"), g("lft"), g("ref_x"), g("ref_logmel"), g("ref_midi"), g("ref_lft")))
        noises.append(d["z"][0].to(dev))
        singles.append(infer(model, d, dev, noise=d["z"]))
    outs = model.inference_ragged(items, noises=noises)
    for (t, _), o, s in zip(shapes, outs, singles):
        assert o.shape == (t, 80) and nerr(o, s) < 2e-5
    # default noise: drawn per item in order on the CPU generator, like a loop of B = 1 calls
    torch.manual_seed(3)
    a = model.inference_ragged(items[:2])
    torch.manual_seed(3)
    b = [model.inference(it[0][None], torch.tensor([it[0].shape[0]]), it[1][None], it[2][None], it[3][None],
                         torch.tensor([it[3].shape[0]]), it[4][None], it[5][None], it[6][None]) for it in items[:2]]
    for o, s in zip(a, b):
        assert nerr(o, s) < 2e-5


def test_rccl_group_of_one(dev, precision):
    """C4 rehearsal on one GPU: bench.py initialises the RCCL ("nccl") process group with device_id before any other
    GPU call and runs its step + waveform gather under it (world size 1, a child process)."""
    if precision != "fp32":
        pytest.skip("one arm is enough: the child process sets its own precision")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-sweep", "--no-train", "--force-dist", "--batch", "2", "--frames", "64",
                        "--ref-frames", "32", "--modes", "fp32"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    mg = out["multi_gpu"]
    assert mg["world_size_seen"] == 1 and mg["backend"] == "nccl" and len(mg["per_rank_ms_per_step"]) == 1
    assert out["value"] > 0 and mg["rank0_gather_ms_per_step"] >= 0
