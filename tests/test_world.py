"""SURVEY 8 row f1 -- the analysis front-end between HiFi-GAN and SiFiGAN (serenade/bin/ssc_postprocessing.py:142-222).

CPU part: the oracle (oracle/world_oracle.py) against the fixtures captured from the reference's in-tree functions
(tests/golden/postproc_f0.npz: `convert_continuos_f0`, the np.interp length match -- PINNED, bit for bit), and, for the
third-party arithmetic (WORLD / SPTK / sifigan: PARITY UNPINNED), properties that do not compare the restatement with
itself: a harmonic source through a known all-pole filter gives that filter's envelope back, a pulse train is periodic
(aperiodicity far below 0 dB) and noise is not, sp2mc -> mc2sp round-trips a smooth envelope, unvoiced frames and F0
jumps stay finite and continuous.  GPU part (-m gpu): every kernel of csrc/world.hip through the C ABI against the
oracle on the same inputs, and the chain decode output -> f1 -> SiFiGAN generator on the card."""
import os

import numpy as np
import pytest
import torch

from oracle import world_oracle as W
from serenade_amd import world

G = os.path.join(os.path.dirname(__file__), "golden")
FS = 24000


# ------------------------------------------------------------------------------------------------ signals
def _allpole():
    def res(fc, bw):
        r = np.exp(-np.pi * bw / FS)
        return np.array([1.0, -2 * r * np.cos(2 * np.pi * fc / FS), r * r])
    a = np.convolve(res(700, 130), res(1800, 200))
    return np.convolve(a, res(3200, 300))


def voiced_signal(seconds=0.5, f0=200.0, seed=0, noise=0.0):
    from scipy.signal import lfilter
    n = int(FS * seconds)
    x = np.zeros(n)
    x[np.arange(0, n, FS / f0).astype(int)] = 1.0
    y = lfilter([1.0], _allpole(), x)
    y /= np.abs(y).max()
    if noise:
        y = y + noise * np.random.default_rng(seed).standard_normal(n)
    return y


def song(seconds=1.0, seed=3):
    """glide + vibrato + an unvoiced stretch + an octave jump; flat harmonic source (tapered out below 10 kHz) through
    the fixed all-pole filter, a little noise: (x, f0 at 5 ms)"""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    n = int(FS * seconds)
    t = np.arange(n) / FS
    f = 180.0 * 2 ** (0.6 * t / seconds) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * t))
    f[t > 0.7 * seconds] *= 2.0
    voiced = ~((t > 0.30 * seconds) & (t < 0.42 * seconds))
    ph = 2 * np.pi * np.cumsum(f) / FS
    src = sum(np.clip((10000.0 - k * f) / 1000.0, 0.0, 1.0) * np.cos(k * ph) for k in range(1, 56))
    x = lfilter([1.0], _allpole(), np.where(voiced, src, 0.0))
    x = 0.5 * x / np.abs(x).max() + (0.0005 + 0.03 * ~voiced) * rng.standard_normal(n)
    tt = W.harvest_time_axis(n, FS)
    idx = np.minimum((tt * FS).astype(int), n - 1)
    f0 = np.where(voiced[idx], f[idx], 0.0)
    return x, f0


# ------------------------------------------------------------------------------------------------ pinned (CPU)
def test_convert_continuos_f0_matches_the_reference_bit_for_bit():
    g = np.load(os.path.join(G, "postproc_f0.npz"))
    for name in ("mixed", "voiced_edges", "all_voiced", "single", "all_zero", "repeated_values"):
        uv, cf0, ok = W.convert_continuos_f0(g[f"{name}_f0"])
        assert ok == bool(g[f"{name}_flag"])
        assert np.array_equal(uv, g[f"{name}_uv"]) and uv.dtype == np.float32
        assert np.array_equal(cf0, g[f"{name}_cf0"]), name


def test_length_match_matches_the_reference_bit_for_bit():
    g = np.load(os.path.join(G, "postproc_f0.npz"))
    for name in ("up", "down", "same"):
        out = W.match_length(g[f"resample_{name}_in"], len(g[f"resample_{name}_out"]))
        ref = g[f"resample_{name}_out"]
        assert np.array_equal(out, ref.squeeze()), name
    assert float(g["alpha_24000"]) == world.ALPHA[24000] == 0.466


# ------------------------------------------------------------------------------------------------ properties (CPU)
def test_sizes_follow_world():
    assert W.cheaptrick_fft_size(FS) == world.cheaptrick_fft_size(FS) == 1024
    assert world._d4c_fft_size(FS) == 2048 and world._n_bands(FS) == 3
    for n in (1, 119, 120, 121, 24000, 245760, 245761):
        assert W.harvest_frame_count(n, FS) == world.harvest_frame_count(n, FS) == n // 120 + 1
    assert np.array_equal(W.harvest_time_axis(2400, FS), world.harvest_time_axis(2400, FS))


def test_cheaptrick_recovers_a_known_envelope():
    from scipy.signal import freqz
    y = voiced_signal()
    t = W.harvest_time_axis(len(y), FS)
    sp = W.cheaptrick(y, np.full(len(t), 200.0), t, FS)
    _, h = freqz([1.0], _allpole(), worN=513, include_nyquist=True)
    env = np.abs(h) ** 2
    band = slice(2, 200)  # 50 Hz .. 4.7 kHz: where the source has harmonics well above the numerical floor
    d = 10 * np.log10(sp[40:60, band] / env[band])
    d -= d.mean(axis=1, keepdims=True)  # the level depends on the source's amplitude, the shape does not
    assert np.sqrt((d ** 2).mean()) < 1.0 and np.abs(d).max() < 5.0   # dB; formants 130-300 Hz wide at F0 = 200 Hz


def test_d4c_separates_a_pulse_train_from_noise():
    y = voiced_signal()
    t = W.harvest_time_axis(len(y), FS)
    f0 = np.full(len(t), 200.0)
    bap = W.d4c_band_aperiodicity(y, f0, t, FS)
    assert (bap[20:80] < -25.0).all() and (bap[20:80, 1:] < -45.0).all()
    noise = 0.1 * np.random.default_rng(1).standard_normal(len(y))
    assert W.d4c_love_train(noise, FS, 200.0, 0.25) < 0.85 < W.d4c_love_train(y, FS, 200.0, 0.25)
    assert np.allclose(W.d4c_band_aperiodicity(noise, f0, t, FS), 0.0, atol=1e-9)      # rejected by the Love Train
    forced = W.d4c_band_aperiodicity(noise, f0, t, FS, threshold=0.0)                   # D4C proper on noise
    assert (forced[20:80] > -6.0).all()
    ap = W.d4c(y, f0, t, FS)
    assert ap.shape == (len(t), 513) and np.allclose(W.code_aperiodicity(ap, FS), bap, atol=1e-9)


def test_sp2mc_round_trip_and_linear_form():
    y = voiced_signal()
    t = W.harvest_time_axis(len(y), FS)
    f0 = np.full(len(t), 200.0)
    sp = W.cheaptrick(y, f0, t, FS)[30:40]
    mc = W.sp2mc(sp, 39, 0.466)
    back = W.mc2sp(mc, 0.466, 1024)
    assert np.abs(10 * np.log10(back / sp)).max() < 1.5      # dB: order-39 mel-cepstrum of a smooth envelope
    full = W.sp2mc(sp, 512, 0.0)                             # alpha 0, full order: plain cepstrum, lossless
    assert np.abs(10 * np.log10(W.mc2sp(full, 0.0, 1024) / sp)).max() < 1e-8
    # the fused route the GPU pipeline takes (liftered cepstrum -> freqt) is the same numbers
    lift = W.cheaptrick(y, f0, t, FS, want="cepstrum")[30:40]
    assert np.abs(W.freqt_from_cepstrum(lift, 39, 0.466) - mc).max() < 1e-12
    # host tables of the product: one matrix per route
    m_sp = world._sp2mc_matrix(torch.device("cpu"), 513, 39, 0.466, False).numpy()
    m_cp = world._sp2mc_matrix(torch.device("cpu"), 513, 39, 0.466, True).numpy()
    assert np.abs(np.log(sp) @ m_sp - mc).max() < 1e-11 and np.abs(lift @ m_cp - mc).max() < 1e-11
    assert np.abs(world._freqt_matrix(40, 12, 0.41) - W.freqt_matrix(40, 12, 0.41)).max() == 0.0


def test_unvoiced_frames_and_f0_jumps_stay_finite_and_continuous():
    x, f0 = song()
    t = W.harvest_time_axis(len(x), FS)
    assert (f0 == 0).any() and (f0 > 0).any()
    sp = W.cheaptrick(x, f0, t, FS)
    assert np.isfinite(sp).all() and (sp > 0).all()
    # F0 = 0 takes WORLD's default-F0 branch: identical to analysing the frame at 500 Hz
    i = int(np.where(f0 == 0)[0][3])
    assert np.array_equal(sp[i], W.cheaptrick_frame(x, FS, 500.0, t[i], 1024))
    # the envelope barely moves across the octave jump of the source (the jump is in F0, not in the filter)
    j = int(np.where((f0[1:] > 1.5 * f0[:-1]) & (f0[:-1] > 0))[0][0])
    mc = W.sp2mc(sp, 39, 0.466)
    assert np.abs(mc[j + 4] - mc[j - 3])[1:].max() < 0.4   # mc[0] (level) follows the number of pulses per window
    bap = W.d4c_band_aperiodicity(x, f0, t, FS)
    assert np.isfinite(bap).all() and (bap <= 0).all()
    assert np.allclose(bap[f0 == 0], 20 * np.log10(1 - 1e-12))
    # digital silence: finite everywhere
    z = np.zeros(2400)
    tz = W.harvest_time_axis(len(z), FS)
    assert np.isfinite(W.cheaptrick(z, np.full(len(tz), 150.0), tz, FS)).all()
    assert np.isfinite(W.d4c_band_aperiodicity(z, np.full(len(tz), 150.0), tz, FS)).all()


def test_excitation_restatement_properties():
    f0 = np.array([0.0, 100.0, 100.0, 200.0, 0.0, 400.0])
    d = W.dilated_factor(f0[:, None].copy(), FS, 4)
    assert np.allclose(d[:, 0], [1.0, 60.0, 60.0, 30.0, 1.0, 15.0])
    s = W.signal_generator_sine(torch.tensor(f0, dtype=torch.float32).view(1, 1, -1), FS, 120, 0.1, 0.0).numpy()[0, 0]
    assert s.shape == (720,) and (s[:120] == 0).all() and (s[480:600] == 0).all()
    ph = np.cumsum(np.repeat(f0 / FS, 120))
    assert np.abs(s - 0.1 * np.sin(2 * np.pi * ph) * np.repeat(f0 > 0, 120)).max() < 1e-5
    q = W.pcm16_roundtrip(np.array([0.0, 1.0, -1.0, 0.5, 1e-5, 2.0]))
    assert np.array_equal(q * 32768, [0, 32767, -32767, 16384, 0, 32767])


# ------------------------------------------------------------------------------------------------ GPU
def _dev():
    return torch.device("cuda:0")


def _cuda(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=_dev(), dtype=dtype)


@pytest.mark.gpu
def test_gpu_pinned_f0_functions_bit_exact():
    g = np.load(os.path.join(G, "postproc_f0.npz"))
    for name in ("mixed", "voiced_edges", "all_voiced", "single", "all_zero", "repeated_values"):
        uv, cf0, ok = world.convert_continuos_f0(_cuda(g[f"{name}_f0"]))
        assert ok == bool(g[f"{name}_flag"]), name
        assert np.array_equal(uv.cpu().numpy(), g[f"{name}_uv"]), name
        assert np.array_equal(cf0.cpu().numpy(), g[f"{name}_cf0"]), name
    names = ("up", "down", "same")
    outs = world.match_length([g[f"resample_{n}_in"] for n in names], [len(g[f"resample_{n}_out"]) for n in names],
                              _dev()).cpu().numpy()
    for i, n in enumerate(names):
        ref = g[f"resample_{n}_out"].squeeze()
        assert np.array_equal(outs[i, :len(ref)], ref), n


@pytest.mark.gpu
def test_gpu_cheaptrick_and_sp2mc_vs_oracle():
    x, f0 = song(1.0)
    t = W.harvest_time_axis(len(x), FS)
    sp_ref = W.cheaptrick(x, f0, t, FS)
    sp = world.cheaptrick(_cuda(x), _cuda(f0), _cuda(t), FS)
    assert sp.shape == sp_ref.shape and sp.dtype == torch.float64
    err = np.abs(np.log(sp.cpu().numpy() / sp_ref)).max()
    # float64 on both sides, yet not 1e-12: WORLD's LinearSmoothing differences a running integral of the power
    # spectrum, so a band 90 dB below the total carries eps * 1e9 -- the oracle itself moves by 1.1e-7 (log) on this
    # signal when its cumsum runs in extended precision.  Summation order is all that differs here.
    assert err < 1e-6, err
    mc_ref = W.sp2mc(sp_ref, 39, 0.466)
    mc = world.sp2mc(sp, 39, 0.466).cpu().numpy()
    assert np.abs(mc - mc_ref).max() < 1e-6
    assert np.abs(world.sp2mc(_cuda(sp_ref), 39, 0.466).cpu().numpy() - mc_ref).max() < 1e-11   # the projection alone
    # batched, other transform sizes
    for fft_size in (512, 2048):
        f0b = np.maximum(f0, 0) if fft_size != 512 else np.where(f0 > 0, np.maximum(f0, 160.0), 0.0)
        ref = W.cheaptrick(x, f0b, t, FS, fft_size=fft_size)
        got = world.cheaptrick(_cuda(np.stack([x, x])), _cuda(np.stack([f0b, f0b])), _cuda(t), FS, fft_size=fft_size)
        assert np.abs(np.log(got[1].cpu().numpy() / ref)).max() < 1e-6
        assert torch.equal(got[0], got[1])


@pytest.mark.gpu
def test_gpu_d4c_vs_oracle():
    x, f0 = song(1.0)
    y = voiced_signal(0.5, noise=0.001)
    for sig, contour in ((x, f0), (y, np.full(W.harvest_frame_count(len(y), FS), 200.0))):
        t = W.harvest_time_axis(len(sig), FS)
        ref = W.d4c_band_aperiodicity(sig, contour, t, FS)
        got = world.d4c_band_aperiodicity(_cuda(sig), _cuda(contour), _cuda(t), FS).cpu().numpy()
        assert got.shape == ref.shape
        # the Love-Train decision is a threshold on a ratio: a frame may only flip if it sits on the threshold
        voiced_ref = ref[:, 0] != ref.max()
        voiced_got = got[:, 0] != got.max()
        assert (voiced_ref == voiced_got).mean() > 0.995
        both = voiced_ref & voiced_got
        assert both.sum() > 20
        assert np.abs(got[both] - ref[both]).max() < 1e-6, np.abs(got[both] - ref[both]).max()   # dB
        assert np.array_equal(got[~voiced_got], np.full((int((~voiced_got).sum()), 3), 20 * np.log10(1 - 1e-12)))


@pytest.mark.gpu
def test_gpu_analyzer_vs_oracle_ragged_batch():
    """the whole stage on a ragged batch of two items: c, cf0, uv, dilated factors, sine (with a given noise draw)"""
    rng = np.random.default_rng(5)
    items = [song(0.8, 3), song(0.55, 4)]
    lf0 = [f0[::2].copy() for _, f0 in items]            # the decode CLI's contour is at 10 ms: half the frames
    waves = [np.clip(x, -0.99, 0.99).astype(np.float32) for x, _ in items]
    n = [len(w) for w in waves]
    buf = np.zeros((2, max(n)), dtype=np.float32)
    for i, w in enumerate(waves):
        buf[i, :n[i]] = w
    mean, scale = rng.standard_normal(43), rng.uniform(0.5, 2.0, 43)

    class S:
        def __init__(self, m, s):
            self.mean_, self.scale_ = m, s
    an = world.Analyzer(scaler={"mcep": S(mean[:40], scale[:40]), "bap": S(mean[40:], scale[40:])})
    F = [W.harvest_frame_count(k, FS) for k in n]
    noise = rng.standard_normal((2, 1, max(F) * 120)).astype(np.float32)
    in_signal, c, dfs, feats = an(_cuda(buf, torch.float32), n, lf0, noise=_cuda(noise, torch.float32))
    assert c.shape == (2, 43, max(F)) and in_signal.shape == (2, 1, max(F) * 120)
    assert [d.shape[-1] for d in dfs] == [max(F) * u for u in (5, 20, 60, 120)]
    for i in range(2):
        ref = W.analyze(W.pcm16_roundtrip(waves[i]), lf0[i], FS, mean=mean, scale=scale)
        assert ref["ok"] and bool(feats["ok"][i].item())
        got_c = c[i, :, :F[i]].T.cpu().numpy()
        unv = 20 * np.log10(1 - 1e-12)
        got_bap = feats["bap"][i, :F[i]].cpu().numpy()
        flips = (got_bap == unv).all(1) != (ref["bap"] == unv).all(1)   # Love-Train threshold decisions
        assert flips.mean() < 0.005
        assert np.abs(got_bap[~flips] - ref["bap"][~flips]).max() < 1e-6
        assert np.abs(feats["mcep"][i, :F[i]].cpu().numpy() - ref["mcep"]).max() < 1e-6
        assert np.abs(got_c[:, :40] - ref["c"][:, :40]).max() < 1e-5                       # float32 output
        assert np.abs(got_c[~flips, 40:] - ref["c"][~flips, 40:]).max() < 1e-4
        assert np.array_equal(feats["cf0"][i, :F[i]].cpu().numpy(), ref["cf0"])              # pinned arithmetic
        assert np.array_equal(feats["uv"][i, :F[i]].cpu().numpy(), ref["uv"])
        for d, dref, u in zip(dfs, ref["dfs"], (5, 20, 60, 120)):
            assert np.array_equal(d[i, 0, :F[i] * u].cpu().numpy(), dref.astype(np.float32))
        cf0_32 = torch.tensor(ref["cf0"], dtype=torch.float32).view(1, 1, -1)
        sref = W.signal_generator_sine(cf0_32, FS, 120, 0.1, 0.003, noise=torch.from_numpy(noise[i:i + 1, :, :F[i] * 120]))
        got = in_signal[i, 0, :F[i] * 120].cpu().numpy()
        assert np.abs(got - sref[0, 0].numpy()).max() < 2e-7
        assert (in_signal[i, 0, F[i] * 120:] == 0).all()


@pytest.mark.gpu
def test_gpu_full_size_properties():
    """the stage at the headline's size (8 items x 10.24 s = 16 392 frames) through properties that need no oracle:
    an item analysed alone equals itself inside the batch bit for bit; doubling the waveform shifts only c0 of the
    mel-cepstrum (by ln 2: the envelope scales by 4) and leaves band aperiodicity, cf0 and uv untouched; everything
    finite, aperiodicities within [-60 dB, 0], unvoiced frames at the unvoiced constant; the sine stays within its
    amplitude"""
    items = [song(10.24, 10 + i) for i in range(8)]
    lf0 = [f0[::2].copy() for _, f0 in items]
    waves = np.stack([np.clip(0.45 * x / np.abs(x).max(), -0.99, 0.99).astype(np.float32) for x, _ in items])
    n = [waves.shape[1]] * 8
    an = world.Analyzer(pcm16=False, noise_amp=0.0)
    wave = _cuda(waves, torch.float32)
    in_signal, c, dfs, feats = an(wave, n, lf0)
    F = W.harvest_frame_count(n[0], FS)
    assert c.shape == (8, 43, F) and 8 * F == 16392
    for t in (in_signal, c, feats["mcep"], feats["bap"], feats["cf0"]):
        assert torch.isfinite(t).all()
    bap = feats["bap"].cpu().numpy()
    unv = 20 * np.log10(1 - 1e-12)
    assert bap.max() <= 0.0 + 1e-9 and bap[bap != unv].min() >= -60.0 - 1e-6
    assert float(in_signal.abs().max()) <= 0.1 + 1e-6                       # sine_amp, no noise given
    # batch independence (item 5 alone)
    _, c1, _, f1 = an(wave[5:6].contiguous(), n[5:6], lf0[5:6])
    assert torch.equal(c1[0], c[5]) and torch.equal(f1["bap"][0], feats["bap"][5])
    assert torch.equal(f1["mcep"][0], feats["mcep"][5]) and torch.equal(f1["cf0"][0], feats["cf0"][5])
    # scale: x -> 2 x
    _, _, _, f2 = an((2.0 * wave).contiguous(), n, lf0)
    d = (f2["mcep"] - feats["mcep"]).cpu().numpy()
    assert np.abs(d[..., 0] - np.log(2.0)).max() < 1e-6 and np.abs(d[..., 1:]).max() < 1e-6
    voiced_same = (f2["bap"] == unv).all(-1) == (feats["bap"] == unv).all(-1)
    assert voiced_same.float().mean() > 0.999                               # Love-Train shares are scale-free
    keep = voiced_same.cpu().numpy()
    assert np.abs(f2["bap"].cpu().numpy()[keep] - bap[keep]).max() < 1e-6
    assert torch.equal(f2["cf0"], feats["cf0"]) and torch.equal(f2["uv"], feats["uv"])


@pytest.mark.gpu
def test_gpu_all_unvoiced_item_is_flagged():
    x = (0.05 * np.random.default_rng(0).standard_normal(4800)).astype(np.float32)
    an = world.Analyzer()
    in_signal, c, dfs, feats = an(_cuda(x[None], torch.float32), [4800], [np.zeros(21)])
    assert int(feats["ok"][0]) == 0 and torch.isfinite(c).all()
    # (a negative contour of another length is clamped at 0 by the length match, as in the reference; at the analysis
    #  length -- 41 frames here -- it reaches the check)
    for bad in ([np.full(21, 7000.0)], [np.full(41, -5.0)], [np.array([])]):
        with pytest.raises(ValueError):
            an(_cuda(x[None], torch.float32), [4800], bad)
    for bad_len in ([0], [4801], [4800, 4800]):
        with pytest.raises(ValueError):
            an(_cuda(x[None], torch.float32), bad_len, [np.zeros(21)])


# ------------------------------------------------------------------------------------------------ stage-9 CLI
def test_postprocessing_cli_overrides():
    from serenade_amd.bin import ssc_postprocessing as PP
    cfg = PP.parse_overrides(["generator=sifigan", "in_dir=/tmp/x", "stats=s.joblib", "checkpoint_path=m.pkl",
                              "noise_amp=0", "dense_factors=[1,2,4,8]", "generator.channels=64"])
    assert cfg["in_dir"] == "/tmp/x" and cfg["noise_amp"] == 0 and cfg["dense_factors"] == [1, 2, 4, 8]
    assert cfg["generator"]["channels"] == 64 and cfg["generator"]["upsample_scales"] == [5, 4, 3, 2]
    assert cfg["sample_rate"] == 24000 and cfg["frame_period"] == 5 and cfg["mcep_dim"] == 39 and cfg["seed"] == 100
    for bad in (["generator=hifigan"], ["nonsense=1"], ["in_dir"], ["generator.nope.x=1"]):
        with pytest.raises(SystemExit):
            PP.parse_overrides(bad)


@pytest.mark.gpu
def test_gpu_postprocessing_cli_end_to_end(tmp_path):
    """decode-CLI outputs (wav + lf0) -> f1 -> SiFiGAN generator on the card, against the CPU restatements chained
    the same way (noise_amp = 0: the reference's noise is a torch.randn draw on the device)."""
    from joblib import dump
    from sklearn.preprocessing import StandardScaler
    from oracle import sifigan_oracle as SO
    from serenade_amd import _shapes, sifigan
    from serenade_amd.bin import ssc_postprocessing as PP
    from serenade_amd.utils.io import read_wav, write_wav_pcm16
    from serenade_amd.utils.synth import fill_state_dict
    from tests._weights import fold_weight_norm
    rng = np.random.default_rng(11)
    d = tmp_path / "results" / "test"
    d.mkdir(parents=True)
    items = {"Alto_song_0001_Breathy": song(0.6, 7), "Tenor_song_0002_Vibrato": song(0.45, 8)}
    for name, (x, f0) in items.items():
        write_wav_pcm16(str(d / f"{name}.wav"), 0.9 * x / np.abs(x).max(), FS)
        np.savez(d / f"{name}.npz", lf0=f0[::2].astype(np.float32))       # 10 ms contour, as ssc_decode writes it
    write_wav_pcm16(str(d / "Alto_song_0001_gt.wav"), items["Alto_song_0001_Breathy"][0] * 0.1, FS)
    write_wav_pcm16(str(d / "00_Breathy_reference.wav"), items["Alto_song_0001_Breathy"][0] * 0.1, FS)
    write_wav_pcm16(str(d / "no_contour.wav"), items["Alto_song_0001_Breathy"][0] * 0.1, FS)
    scaler = {"mcep": StandardScaler().fit(rng.standard_normal((60, 40)) * 2 - 1),
              "bap": StandardScaler().fit(rng.standard_normal((60, 3)) * 10 - 20)}
    dump(scaler, tmp_path / "stats.joblib")
    cfg = sifigan.DEFAULT_PARAMS
    sd = fill_state_dict(_shapes.as_meta(sifigan.sifigan_shapes(**cfg)), seed=5)
    torch.save({"model": {"generator": sd}}, tmp_path / "model.pkl")
    import serenade_amd
    serenade_amd.set_precision("fp32")
    try:
        frames = PP.main(["generator=sifigan", f"in_dir={tmp_path / 'results'}", f"stats={tmp_path / 'stats.joblib'}",
                          f"checkpoint_path={tmp_path / 'model.pkl'}", "noise_amp=0"])
    finally:
        serenade_amd.set_precision("fp32")  # the package default
    names = sorted(p.name for p in d.iterdir() if p.name.endswith("_sifigan.wav"))
    assert names == [f"{n}_sifigan.wav" for n in sorted(items)]
    w = fold_weight_norm(sd)
    mean = np.concatenate([scaler["mcep"].mean_, scaler["bap"].mean_])
    scale = np.concatenate([scaler["mcep"].scale_, scaler["bap"].scale_])
    total = 0
    for name in items:
        x, sr = read_wav(str(d / f"{name}.wav"))
        ref = W.analyze(x, np.load(d / f"{name}.npz")["lf0"], FS, mean=mean, scale=scale)
        total += len(ref["f0"])
        cf0_32 = torch.tensor(ref["cf0"], dtype=torch.float32).view(1, 1, -1)
        sine = W.signal_generator_sine(cf0_32, FS, 120, 0.1, 0.0)
        c = torch.tensor(ref["c"], dtype=torch.float32).T.unsqueeze(0)
        dfs = [torch.tensor(v, dtype=torch.float32).view(1, 1, -1) for v in ref["dfs"]]
        y_ref, _ = SO.sifigan_forward(w, sine, c, dfs, cfg)
        y, sr2 = read_wav(str(d / f"{name}_sifigan.wav"))
        assert sr2 == FS and len(y) == y_ref.numel() == len(ref["f0"]) * 120
        q = np.clip(np.rint(y_ref.view(-1).numpy().astype(np.float64) * 32767.0), -32768, 32767) / 32768.0
        assert np.abs(y - q).max() <= 2.0 / 32768.0      # PCM_16 LSBs: fp32 generator on both sides
    assert frames == total
