"""serenade-decode mirror (CPU, through the C-ABI emulator): command line, file formats, outputs; and the
F0 helpers of the decode loop against hand-computed values (reference: serenade/bin/ssc_decode.py:32-154,190-455)."""
import json
import os
import wave as wavmod

import numpy as np
import pytest
import torch
import yaml

from oracle import serenade_oracle as O
from serenade_amd import _shapes, models, vocoder
from serenade_amd.bin import ssc_decode
from serenade_amd.utils import f0 as F0
from serenade_amd.utils.synth import fill_state_dict
from tests import _emulator
from tests._weights import fold_weight_norm

MODEL_PARAMS = dict(input_dim=32, output_dim=80, encoder_channels=80, decoder_channels=256, gst_embed_dim=256,
                    decoder_attention_head_dim=64)
GEN_PARAMS = dict(in_channels=80, out_channels=1, channels=64, kernel_size=7, upsample_scales=[4, 2],
                  upsample_kernel_sizes=[8, 4], resblock_kernel_sizes=[3, 5], resblock_dilations=[[1, 2], [2, 6, 3]],
                  use_additional_convs=True, bias=True, nonlinear_activation="LeakyReLU",
                  nonlinear_activation_params={"negative_slope": 0.1}, use_causal_conv=False, use_weight_norm=True)


def test_f0_helpers_known_answers():
    assert abs(F0.C4_HZ - 261.6255653) < 1e-6
    hz = np.array([0.0, F0.C4_HZ, 2 * F0.C4_HZ, 440.0])
    cent = F0.hz_to_cent_based_c4(hz)
    assert np.allclose(cent, [0.0, 4800.0, 6000.0, 5700.0])
    assert np.allclose(F0.cent_to_hz_based_c4(cent), hz)
    st = F0.F0Statistics()
    a = np.array([0.0, 100.0, 200.0, 0.0, 400.0])
    m, s = st.estimate([a])
    assert np.isclose(m, np.log(200.0)) and np.isclose(s, np.std(np.log([100.0, 200.0, 400.0])))
    assert np.allclose(st.convert(a, [m, s], [m + np.log(2), s]), 2 * a)
    # +7 semitones between the mean pitches -> scaled by 1.4 -> rounded to +1000 cents
    src = np.array([0.0, 220.0, 220.0, 0.0])
    trg = np.array([220.0 * 2 ** (7 / 12)] * 3)
    out = F0.linear_midi_shift(src.copy(), trg)
    assert np.allclose(out[[1, 2]], 220.0 * 2 ** (10 / 12)) and out[0] == 0 and out[3] == 0
    # -7 semitones -> scaled by 5/7 -> -500 cents
    out = F0.linear_midi_shift(src.copy(), np.array([220.0 * 2 ** (-7 / 12)] * 3))
    assert np.allclose(out[[1, 2]], 220.0 * 2 ** (-5 / 12))


def _feats(rng, T):
    return dict(wave=(0.1 * rng.standard_normal(T * 240)).astype(np.float32),
                hubert=rng.standard_normal((T, 32)).astype(np.float32),
                logmel=rng.standard_normal((T, 80)).astype(np.float32),
                loud=rng.uniform(-40, 0, (T, 1)).astype(np.float32),
                est_lf0_score=rng.uniform(40, 70, (T, 1)).astype(np.float32),
                midi=rng.uniform(40, 70, (T, 1)).astype(np.float32),
                f0=np.where(rng.uniform(size=T) > 0.2, rng.uniform(150, 400, T), 0.0))


def _cli_case(tmp_path, on_gpu):
    """Build a tiny dump / checkpoint / vocoder / stats tree, run the CLI on it (through the C-ABI emulator on the
    CPU, or the real library on the GPU) and check files, formats and values against the oracle."""
    import contextlib
    from joblib import dump
    from sklearn.preprocessing import MinMaxScaler, StandardScaler
    rng = np.random.default_rng(0)
    dumpdir, outdir, exp = tmp_path / "dump", tmp_path / "out", tmp_path / "exp"
    for d in (dumpdir, exp, tmp_path / "ref", tmp_path / "voc"):
        d.mkdir()
    src, ref = _feats(rng, 24), _feats(rng, 16)
    np.savez(dumpdir / "EN_spk1_song_Control_Group_0001.npz", **src)
    np.savez(tmp_path / "ref" / "breathy.npz", **ref)
    scaler = {"logmel": StandardScaler().fit(rng.standard_normal((50, 80))),
              "hubert": StandardScaler().fit(rng.standard_normal((50, 32))),
              "score": MinMaxScaler().fit(np.array([[30.0], [80.0]])),
              "loud": MinMaxScaler().fit(np.array([[-50.0], [0.0]]))}
    dump(scaler, tmp_path / "stats.joblib")
    sd = fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**MODEL_PARAMS)), seed=3)
    torch.save({"model": sd}, exp / "checkpoint.pkl")
    gsd = fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**GEN_PARAMS, weight_norm=True)), seed=4)
    torch.save({"model": {"generator": gsd}}, tmp_path / "voc" / "vocoder.pkl")
    yaml.safe_dump({"generator_params": GEN_PARAMS, "sampling_rate": 24000, "format": "hdf5"},
                   open(tmp_path / "voc" / "config.yml", "w"))
    vstats = {"mean": 0.1 * rng.standard_normal(80).astype(np.float32),
              "scale": (1 + 0.1 * rng.uniform(size=80)).astype(np.float32)}
    np.savez(tmp_path / "voc" / "stats.npz", **vstats)
    yaml.safe_dump({"model_type": "Serenade", "model_params": MODEL_PARAMS, "sampling_rate": 24000,
                    "vocoder": {"checkpoint": str(tmp_path / "voc" / "vocoder.pkl"),
                                "config": str(tmp_path / "voc" / "config.yml"),
                                "stats": str(tmp_path / "voc" / "stats.npz")}},
                   open(exp / "config.yml", "w"))
    json.dump({"Breathy": str(tmp_path / "ref" / "breathy.npz")}, open(tmp_path / "refs.json", "w"))

    with (contextlib.nullcontext() if on_gpu else _emulator.installed()):
        torch.manual_seed(123)
        ssc_decode.main(["--dumpdir", str(dumpdir), "--stats", str(tmp_path / "stats.joblib"), "--ref-dict",
                         str(tmp_path / "refs.json"), "--outdir", str(outdir), "--checkpoint",
                         str(exp / "checkpoint.pkl"), "--verbose", "0"])
    utt = "EN_spk1_song_Control_Group_0001"
    names = sorted(os.listdir(outdir))
    assert names == sorted([f"{utt}_gt.wav", "00_Breathy_reference.wav", f"{utt}_Breathy.wav", f"{utt}_Breathy.npz"])
    with wavmod.open(str(outdir / f"{utt}_Breathy.wav")) as f:
        assert (f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()) == (24000, 1, 2, 24 * 8)  # hop of GEN_PARAMS = 4 * 2
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    lf0 = np.load(outdir / f"{utt}_Breathy.npz")["lf0"]
    assert lf0.dtype == np.float32 and lf0.shape == (24,)
    assert np.allclose(lf0, F0.linear_midi_shift(src["f0"].copy(), ref["f0"]).astype(np.float32))

    # the same conversion through the oracle (identical CPU-generator noise)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    x = t((src["hubert"] - scaler["hubert"].mean_) / scaler["hubert"].scale_).unsqueeze(0)
    score = t((src["est_lf0_score"] - 30.0) / 50.0).unsqueeze(0)
    loud = t((src["loud"] + 50.0) / 50.0).unsqueeze(0)
    rx = t((ref["hubert"] - scaler["hubert"].mean_) / scaler["hubert"].scale_).unsqueeze(0)
    rmel = t((ref["logmel"] - scaler["logmel"].mean_) / scaler["logmel"].scale_).unsqueeze(0)
    rscore = t((ref["est_lf0_score"] - 30.0) / 50.0).view(1, -1, 1)
    rloud = t((ref["loud"] + 50.0) / 50.0).unsqueeze(0)
    torch.manual_seed(123)
    z = torch.randn((1, 80, 40)) * 0.667
    mel = O.serenade_inference(sd, x, torch.tensor([24]), score, loud, rx, torch.tensor([16]), rmel, rscore, rloud, z)
    trg = {"mean": t(scaler["logmel"].mean_), "scale": t(scaler["logmel"].scale_)}
    wav = O.vocoder_decode(fold_weight_norm(gsd), mel, GEN_PARAMS, {k: t(v) for k, v in vstats.items()}, trg)
    ref_pcm = np.rint(wav.double().numpy() * 32767.0).astype(np.int16)  # libsndfile: lrint(x * 0x7FFF)
    assert np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32)).max() <= 1


def test_decode_cli_end_to_end(tmp_path):
    _cli_case(tmp_path, on_gpu=False)


def _two_style_tree(tmp_path):
    """dump with one utterance and two prompts of different lengths; returns the CLI argument list (no --outdir)"""
    from joblib import dump
    from sklearn.preprocessing import MinMaxScaler, StandardScaler
    rng = np.random.default_rng(1)
    for d in ("dump", "exp", "ref", "voc"):
        (tmp_path / d).mkdir()
    np.savez(tmp_path / "dump" / "EN_spk1_song_Control_Group_0001.npz", **_feats(rng, 20))
    np.savez(tmp_path / "dump" / "EN_spk1_song_Control_Group_0002.npz", **_feats(rng, 27))
    np.savez(tmp_path / "ref" / "breathy.npz", **_feats(rng, 13))
    np.savez(tmp_path / "ref" / "falsetto.npz", **_feats(rng, 18))
    scaler = {"logmel": StandardScaler().fit(rng.standard_normal((50, 80))),
              "hubert": StandardScaler().fit(rng.standard_normal((50, 32))),
              "score": MinMaxScaler().fit(np.array([[30.0], [80.0]])),
              "loud": MinMaxScaler().fit(np.array([[-50.0], [0.0]]))}
    dump(scaler, tmp_path / "stats.joblib")
    sd = fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**MODEL_PARAMS)), seed=3)
    torch.save({"model": sd}, tmp_path / "exp" / "checkpoint.pkl")
    gsd = fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**GEN_PARAMS, weight_norm=True)), seed=4)
    torch.save({"model": {"generator": gsd}}, tmp_path / "voc" / "vocoder.pkl")
    yaml.safe_dump({"generator_params": GEN_PARAMS, "sampling_rate": 24000, "format": "hdf5"},
                   open(tmp_path / "voc" / "config.yml", "w"))
    np.savez(tmp_path / "voc" / "stats.npz", mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
    yaml.safe_dump({"model_type": "Serenade", "model_params": MODEL_PARAMS, "sampling_rate": 24000,
                    "vocoder": {"checkpoint": str(tmp_path / "voc" / "vocoder.pkl"),
                                "config": str(tmp_path / "voc" / "config.yml"),
                                "stats": str(tmp_path / "voc" / "stats.npz")}},
                   open(tmp_path / "exp" / "config.yml", "w"))
    json.dump({"Breathy": str(tmp_path / "ref" / "breathy.npz"), "Falsetto": str(tmp_path / "ref" / "falsetto.npz")},
              open(tmp_path / "refs.json", "w"))
    return ["--dumpdir", str(tmp_path / "dump"), "--stats", str(tmp_path / "stats.joblib"), "--ref-dict",
            str(tmp_path / "refs.json"), "--checkpoint", str(tmp_path / "exp" / "checkpoint.pkl"), "--verbose", "0"]


def _batch_styles_case(tmp_path, on_gpu):
    import contextlib
    argv = _two_style_tree(tmp_path)
    outs = {}
    with (contextlib.nullcontext() if on_gpu else _emulator.installed()):
        for name, extra in (("loop", []), ("batch", ["--batch-styles"]), ("utts", ["--batch-utterances", "2"])):
            torch.manual_seed(7)
            ssc_decode.main(argv + ["--outdir", str(tmp_path / name)] + extra)
            outs[name] = sorted(os.listdir(tmp_path / name))
    # 2 utterances x (gt + 2 x (wav, lf0)) + 2 reference wavs
    assert outs["loop"] == outs["batch"] == outs["utts"] and len(outs["loop"]) == 12
    for f, other in [(f, o) for f in outs["loop"] for o in ("batch", "utts")]:
        a, b = tmp_path / "loop" / f, tmp_path / other / f
        if f.endswith(".wav"):
            with wavmod.open(str(a)) as fa, wavmod.open(str(b)) as fb:
                pa = np.frombuffer(fa.readframes(fa.getnframes()), dtype="<i2").astype(np.int32)
                pb = np.frombuffer(fb.readframes(fb.getnframes()), dtype="<i2").astype(np.int32)
            assert pa.shape == pb.shape and np.abs(pa - pb).max() <= 1, f
        else:
            assert np.array_equal(np.load(a)["lf0"], np.load(b)["lf0"]), f


def test_batch_styles_equals_the_style_loop(tmp_path):
    """--batch-styles: one exact ragged batch per utterance (different prompt lengths) == the B = 1 style loop"""
    _batch_styles_case(tmp_path, on_gpu=False)


@pytest.mark.gpu
def test_batch_styles_equals_the_style_loop_gpu(tmp_path):
    assert torch.cuda.is_available()
    _batch_styles_case(tmp_path, on_gpu=True)


@pytest.mark.gpu
def test_decode_cli_end_to_end_gpu(tmp_path):
    """the same case through libserenade_hip.so on cuda:0 (B = 1 loop, files, PCM_16, lf0) vs the CPU oracle"""
    assert torch.cuda.is_available()
    _cli_case(tmp_path, on_gpu=True)


@pytest.mark.gpu
def test_decode_cli_in_process_stage9_equals_the_two_cli_chain(tmp_path):
    """--sifigan-checkpoint: mel -> HiFi-GAN -> WORLD analysis -> SiFiGAN in one process on the GPU writes the samples
    the file-based chain (serenade-decode, then serenade-postprocessing on its wav + lf0 files) writes"""
    from joblib import dump
    from sklearn.preprocessing import StandardScaler
    from serenade_amd import sifigan
    from serenade_amd.bin import ssc_postprocessing
    argv = _two_style_tree(tmp_path)
    rng = np.random.default_rng(3)
    dump({"mcep": StandardScaler().fit(rng.standard_normal((40, 40)) - 2), "bap": StandardScaler().fit(rng.standard_normal((40, 3)) * 9 - 10)},
         tmp_path / "sifigan_stats.joblib")
    gsd = fill_state_dict(_shapes.as_meta(sifigan.sifigan_shapes(**sifigan.DEFAULT_PARAMS)), seed=6)
    torch.save({"model": {"generator": gsd}}, tmp_path / "sifigan.pkl")
    torch.manual_seed(7)
    ssc_decode.main(argv + ["--outdir", str(tmp_path / "one"), "--sifigan-checkpoint", str(tmp_path / "sifigan.pkl"),
                            "--sifigan-stats", str(tmp_path / "sifigan_stats.joblib"), "--sifigan-noise-amp", "0"])
    torch.manual_seed(7)
    ssc_decode.main(argv + ["--outdir", str(tmp_path / "two")])
    ssc_postprocessing.main(["generator=sifigan", f"in_dir={tmp_path / 'two'}", f"stats={tmp_path / 'sifigan_stats.joblib'}",
                             f"checkpoint_path={tmp_path / 'sifigan.pkl'}", "noise_amp=0"])
    names = sorted(n for n in os.listdir(tmp_path / "one") if n.endswith("_sifigan.wav"))
    assert len(names) == 4 and names == sorted(n for n in os.listdir(tmp_path / "two") if n.endswith("_sifigan.wav"))
    for n in names:
        with wavmod.open(str(tmp_path / "one" / n)) as fa, wavmod.open(str(tmp_path / "two" / n)) as fb:
            pa = np.frombuffer(fa.readframes(fa.getnframes()), dtype="<i2").astype(np.int32)
            pb = np.frombuffer(fb.readframes(fb.getnframes()), dtype="<i2").astype(np.int32)
        assert pa.shape == pb.shape and pa.size > 0 and np.abs(pa - pb).max() <= 1, n


def test_cli_rejects_bad_argument_combinations(tmp_path):
    with pytest.raises(SystemExit):
        ssc_decode.main(["--outdir", str(tmp_path)])  # --stats / --checkpoint are required


# ---- rank sharding of the decode loop: two gloo ranks, every conversion exactly once
def _sharded_worker(rank, world, port, root, q):
    import torch.distributed as dist
    from pathlib import Path
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)  # two workers on the CI box's 8 cores
    written = []
    real = ssc_decode.write_wav_pcm16
    ssc_decode.write_wav_pcm16 = lambda path, wave, sr: (written.append(os.path.basename(path)), real(path, wave, sr))[1]
    root = Path(root)
    argv = ["--dumpdir", str(root / "dump"), "--stats", str(root / "stats.joblib"), "--ref-dict", str(root / "refs.json"),
            "--checkpoint", str(root / "exp" / "checkpoint.pkl"), "--verbose", "0", "--outdir", str(root / "out")]
    with _emulator.installed():
        torch.manual_seed(7)
        ssc_decode.main(argv)
        err = None
        try:  # no --ref-dict under several ranks: every rank would draw its own prompts
            ssc_decode.main([a for a in argv if a not in ("--ref-dict", str(root / "refs.json"))])
        except ValueError as e:
            err = str(e)
    q.put((rank, written, err))
    dist.barrier()
    dist.destroy_process_group()


def test_decode_cli_shards_utterances_over_two_gloo_ranks(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_parallel_gloo import _free_port
    _two_style_tree(tmp_path)
    rng = np.random.default_rng(9)
    for i in (3, 4, 5):  # 5 utterances over 2 ranks: 3 + 2
        np.savez(tmp_path / "dump" / f"EN_spk1_song_Control_Group_000{i}.npz", **_feats(rng, 14 + i))
    (tmp_path / "out").mkdir()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (w, e)) for r, w, e in (q.get() for _ in procs))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    utts = [f"EN_spk1_song_Control_Group_000{i}" for i in range(1, 6)]
    conv = lambda names: sorted(n for n in names if not n.startswith("00_") and not n.endswith("_gt.wav"))
    per_rank = {r: conv(w) for r, (w, _) in got.items()}
    assert per_rank[0] == sorted(f"{u}_{s}.wav" for u in utts[:3] for s in ("Breathy", "Falsetto"))
    assert per_rank[1] == sorted(f"{u}_{s}.wav" for u in utts[3:] for s in ("Breathy", "Falsetto"))
    everything = got[0][0] + got[1][0]
    assert len(everything) == len(set(everything))          # nothing written twice, by either rank
    assert sorted(n for n in everything if n.startswith("00_")) == ["00_Breathy_reference.wav", "00_Falsetto_reference.wav"]
    assert all(n in got[0][0] for n in ("00_Breathy_reference.wav", "00_Falsetto_reference.wav"))  # rank 0 only
    files = sorted(os.listdir(tmp_path / "out"))
    assert len(files) == 5 * (1 + 2 * 2) + 2
    assert all("several ranks need --ref-dict" in e for _, e in got.values())
