"""Feature front-end (reference: serenade/bin/preprocess.py:126-203).

PARITY UNPINNED: the reference computes these features with librosa, which is neither vendored nor installed here, and
holds no fixture for them.  So (1) the oracle's restatement of librosa (oracle/features_oracle.py) is checked against
INDEPENDENT formulas -- a brute-force DFT, scipy's window, analytic properties of the Slaney filterbank and of the
A-curve -- and (2) the product (serenade_amd.features over the C ABI) is checked against the oracle: on the CPU through
the ABI emulator (host logic: padding, signal view, tap groups, tables) and on the GPU for real.
"""
import numpy as np
import pytest
import torch

from oracle import features_oracle as FO
from serenade_amd import features
from tests import _emulator

SR, FFT, WIN, HOP, MELS, FMIN, FMAX = 24000, 512, 480, 240, 80, 63, 12000  # conf/serenade.yaml:4-21


def _audio(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    x = 0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 3300.0 * t + 1.0) + 0.02 * rng.standard_normal(n)
    x[: n // 7] *= 1e-3  # a near-silent stretch: exercises eps / amin / the top_db floor
    return x.astype(np.float32)


# ------------------------------------------------------------------ the oracle against independent formulas
def test_oracle_stft_equals_a_direct_dft_of_reflect_padded_frames():
    import scipy.signal
    x = _audio(2000).astype(np.float64)
    S = FO.stft(x, FFT, HOP, WIN)
    assert S.shape == (257, 1 + 2000 // HOP)
    w = np.zeros(FFT)
    w[16:496] = scipy.signal.get_window("hann", WIN, fftbins=True)
    assert np.allclose(w[16:496], FO.hann_periodic(WIN), atol=1e-15)
    xp = np.concatenate([x[1:257][::-1], x, x[-257:-1][::-1]])  # reflect without repeating the edge sample
    k = np.arange(FFT)
    for fr in (0, 3, 8):
        seg = xp[fr * HOP: fr * HOP + FFT] * w
        for f in (0, 1, 100, 256):
            assert abs(np.sum(seg * np.exp(-2j * np.pi * f * k / FFT)) - S[f, fr]) < 1e-9


def test_oracle_slaney_filterbank_properties():
    fb = FO.mel_filterbank(SR, FFT, MELS, FMIN, FMAX)
    assert fb.shape == (80, 257) and (fb >= 0).all()
    # Slaney scale: linear below 1 kHz (200/3 Hz per mel), log above with 27 steps per factor 6.4
    assert np.isclose(FO.hz_to_mel(1000.0), 15.0) and np.isclose(FO.hz_to_mel(6400.0), 15.0 + 27.0)
    assert np.allclose(FO.mel_to_hz(FO.hz_to_mel(np.array([63.0, 500.0, 1000.0, 7000.0]))), [63.0, 500.0, 1000.0, 7000.0])
    edges = FO.mel_to_hz(np.linspace(FO.hz_to_mel(FMIN), FO.hz_to_mel(FMAX), MELS + 2))
    df = SR / FFT
    for i in (5, 40, 79):
        on = np.nonzero(fb[i])[0] * df
        assert on.min() > edges[i] - 1e-9 and on.max() < edges[i + 2] + 1e-9  # support = (lower edge, upper edge)
        # area normalisation: a triangle of height 2 / width has unit area; sampled on the FFT grid within ~1 bin
        assert abs(fb[i].sum() * df - 1.0) < 0.5 * df * fb[i].max() + 0.05


def test_oracle_a_weighting_anchor_points():
    a = FO.a_weighting(np.array([0.0, 100.0, 1000.0, 10000.0]))
    assert a[0] == -80.0                       # clipped
    assert abs(a[2]) < 0.01                    # 0 dB at 1 kHz by definition
    assert abs(a[1] - (-19.1)) < 0.1 and abs(a[3] - (-2.5)) < 0.1  # IEC 61672 table values


def test_oracle_loudness_of_a_scaled_signal_shifts_by_the_gain():
    x = _audio(6000, seed=3).astype(np.float64)
    a, b = FO.loudness_extract(x, SR, HOP), FO.loudness_extract(2.0 * x, SR, HOP)
    # amplitude doubles -> +6.02 dB everywhere, the top_db floor moves with the maximum -> mean amplitude doubles
    assert np.allclose(np.exp(b) - 1e-5, 2.0 * (np.exp(a) - 1e-5), rtol=1e-9)


# ------------------------------------------------------------------ product vs oracle
def _check(dev, n, B):
    xs = np.stack([_audio(n, seed=s) for s in range(B)])
    x = torch.from_numpy(xs).to(dev)
    mel = features.logmelfilterbank(x if B > 1 else x[0], SR, fft_size=FFT, hop_size=HOP, win_length=WIN, window="hann",
                                    num_mels=MELS, fmin=FMIN, fmax=FMAX)
    loud = features.loudness_extract(x if B > 1 else x[0], SR, HOP)
    loud_r = features.loudness_extract(x if B > 1 else x[0], SR, HOP, pad_mode="reflect")
    frames = 1 + n // HOP
    assert mel.shape == ((B, frames, MELS) if B > 1 else (frames, MELS))
    assert loud.shape == ((B, frames) if B > 1 else (frames,))
    mel, loud = mel.reshape(B, frames, MELS).cpu().numpy(), loud.reshape(B, frames).cpu().numpy()
    for b in range(B):
        ref = FO.logmelfilterbank(xs[b], SR, fft_size=FFT, hop_size=HOP, win_length=WIN, num_mels=MELS, fmin=FMIN,
                                  fmax=FMAX)
        assert np.abs(mel[b] - ref).max() < 2e-4  # log10 units; fp32 STFT + fp32 log
        assert np.abs(loud[b] - FO.loudness_extract(xs[b], SR, HOP)).max() < 2e-4
        ref_r = FO.loudness_extract(xs[b], SR, HOP, pad_mode="reflect")  # librosa < 0.10's default for this call
        assert np.abs(loud_r.reshape(B, frames).cpu().numpy()[b] - ref_r).max() < 2e-4
        k = 2048 // 2 // HOP + 1  # only frames whose window reaches over an end see the padding
        if frames > 2 * k:
            assert np.abs(FO.loudness_extract(xs[b], SR, HOP)[k:-k] - ref_r[k:-k]).max() < 1e-12
    # natural log / base 2 variants and the default (win_length = fft_size, full band) go through the same code
    m2 = features.logmelfilterbank(x[0], SR, fft_size=FFT, hop_size=HOP, log_base=None)
    ref = FO.logmelfilterbank(xs[0], SR, fft_size=FFT, hop_size=HOP, log_base=None)
    assert np.abs(m2.cpu().numpy() - ref).max() < 5e-4


def test_front_end_host_logic_through_the_emulator():
    with _emulator.installed():
        features._PLANS.clear()
        _check(torch.device("cpu"), 2400 + 17, 2)
        _check(torch.device("cpu"), 1100, 1)
        with pytest.raises(ValueError):
            features.logmelfilterbank(torch.zeros(1000), SR, fft_size=510, hop_size=255)
        with pytest.raises(ValueError):
            features.logmelfilterbank(torch.zeros(1000), SR, fft_size=512, hop_size=240, log_base=3.0)
    features._PLANS.clear()


@pytest.mark.gpu
def test_front_end_on_the_gpu():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    features._PLANS.clear()
    _check(dev, 24000 * 3 + 101, 3)   # 3 s clips, ragged tail
    _check(dev, 2000, 1)
    with pytest.raises(RuntimeError):
        features.logmelfilterbank(torch.zeros(1000), SR, fft_size=FFT, hop_size=HOP)  # CPU tensor: no fallback


@pytest.mark.gpu
def test_front_end_at_utterance_scale_against_the_restatement(capsys):
    """8 utterances of 1024 frames (10.24 s): values against the numpy restatement on one utterance, and the
    restatement's host time beside the GPU's (printed: the f3 side-by-side, tools/featbench.py times the GPU alone)"""
    import time
    assert torch.cuda.is_available()
    B, T = 8, 1024
    n = T * HOP
    audio = (np.random.default_rng(0).standard_normal((B, n)) * 0.1).astype(np.float32)
    a = torch.from_numpy(audio).cuda()
    kw = dict(fft_size=FFT, hop_size=HOP, win_length=480, num_mels=80, fmin=63, fmax=12000)
    report = {}
    for name, gpu_fn, cpu_fn, tol in (
            ("logmel", lambda: features.logmelfilterbank(a, SR, **kw), lambda: FO.logmelfilterbank(audio[0], SR, **kw), 5e-5),
            ("loudness", lambda: features.loudness_extract(a, SR, HOP), lambda: FO.loudness_extract(audio[0], SR, HOP), 5e-6)):
        for _ in range(3):
            r = gpu_fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = gpu_fn()
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / 10
        t1 = time.perf_counter()
        ref = cpu_fn()
        cpu = time.perf_counter() - t1
        err = float(np.abs(r[0].cpu().numpy()[: len(ref)] - ref).max())
        assert err < tol, (name, err)
        report[name] = {"gpu_frames_per_s": B * r.shape[1] / gpu, "cpu_numpy_frames_per_s": len(ref) / cpu, "max_abs_err": err}
        assert report[name]["gpu_frames_per_s"] > 10 * report[name]["cpu_numpy_frames_per_s"]
    with capsys.disabled():
        print("\nfront-end side-by-side:", report)
