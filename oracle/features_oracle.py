"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, float64) restatement of the feature front-end in front of the hot path
(SURVEY.md section 8f rank 3): `logmelfilterbank` and `loudness_extract` of serenade/bin/preprocess.py:126-203, with the
recipe's analysis settings (egs/gtsinger/ssc1/conf/serenade.yaml:4-21: 24 kHz, fft 512, window 480 hann, hop 240, 80 mels
63-12000 Hz, log10, eps 1e-10).

**PARITY UNPINNED.**  The reference delegates all of this arithmetic to `librosa` (un-vendored; `setup.cfg` lists it
without a pin), which is not installed in the build container, and the reference holds no fixture for these functions.
What follows restates librosa's published behaviour (0.10.x: `stft` with center padding -- pad_mode "reflect" where the
call site passes it (logmelfilterbank, preprocess.py:180), the library default elsewhere (loudness_extract, :131):
"constant" zeros since 0.10, "reflect" before -- and a periodic Hann window zero-padded to n_fft, `filters.mel` with the Slaney scale and Slaney area normalisation, `power_to_db` with
ref 1.0 / amin 1e-10 / top_db 80, `A_weighting`, `db_to_amplitude`) anchored on the reference's call sites
preprocess.py:130-137 and :176-199.  It is cross-checked in tests/ against independent formulas (a direct DFT,
scipy's window, the filterbank's partition-of-unity / area properties), never against librosa itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import numpy as np


# ------------------------------------------------------------------ librosa.stft (center=True, pad_mode="reflect")
def hann_periodic(win_length):
    """scipy.signal.get_window("hann", M, fftbins=True)"""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def stft(audio, n_fft, hop_length, win_length=None, pad_mode="reflect"):
    """(1 + n_fft // 2, 1 + len(audio) // hop_length) complex128"""
    win_length = n_fft if win_length is None else win_length
    w = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2  # util.pad_center
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    y = np.pad(np.asarray(audio, dtype=np.float64), n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    return np.fft.rfft(y[idx] * w[None, :], axis=1).T


# ------------------------------------------------------------------ librosa.filters.mel (htk=False, norm="slaney")
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= _MIN_LOG_HZ, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP, f / _F_SP)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _MIN_LOG_MEL, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), _F_SP * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """(n_mels, 1 + n_fft // 2) float64"""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, log_base=10.0):
    """preprocess.py:140-203 -> (#frames, num_mels)"""
    assert window == "hann"
    spc = np.abs(stft(audio, fft_size, hop_size, win_length)).T
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    mel = np.maximum(eps, spc @ mel_filterbank(sampling_rate, fft_size, num_mels, fmin, fmax).T)
    if log_base is None:
        return np.log(mel)
    if log_base == 10.0:
        return np.log10(mel)
    if log_base == 2.0:
        return np.log2(mel)
    raise ValueError(f"{log_base} is not supported.")


# ------------------------------------------------------------------ loudness (A-weighted), preprocess.py:126-137
def a_weighting(frequencies, min_db=-80.0):
    f_sq = np.asarray(frequencies, dtype=np.float64) ** 2
    c = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2
    with np.errstate(divide="ignore"):
        w = 2.0 + 20.0 * (np.log10(c[0]) + 2 * np.log10(f_sq) - np.log10(f_sq + c[0]) - np.log10(f_sq + c[1])
                          - 0.5 * np.log10(f_sq + c[2]) - 0.5 * np.log10(f_sq + c[3]))
    return np.maximum(min_db, w)


def loudness_extract(audio, sampling_rate, hop_length, n_fft=2048, pad_mode="constant"):
    """-> (#frames,) log mean A-weighted amplitude.  librosa.stft / fft_frequencies defaults: n_fft = 2048 and, since
    librosa 0.10, zero ("constant") padding of the centred frames; pad_mode="reflect" is the pre-0.10 default."""
    power = np.abs(stft(audio, n_fft, hop_length, pad_mode=pad_mode)) ** 2  # (bins, frames)
    bins = np.linspace(0.0, sampling_rate / 2.0, 1 + n_fft // 2)
    db = 10.0 * np.log10(np.maximum(1e-10, power))       # power_to_db(ref=1.0, amin=1e-10, top_db=80.0)
    db = np.maximum(db, db.max() - 80.0)
    loud = a_weighting(bins)[:, None] + db                # perceptual_weighting
    amp = np.power(10.0, 0.05 * loud)                     # db_to_amplitude
    return np.log(np.mean(amp, axis=0) + 1e-5)
