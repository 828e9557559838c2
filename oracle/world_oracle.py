"""CPU ORACLE for SURVEY 8 row f1 (the analysis front-end between HiFi-GAN and SiFiGAN) -- test infrastructure, NOT
product code.  numpy float64, one python loop over frames: sized for seconds of audio.

What the reference runs (serenade/bin/ssc_postprocessing.py:142-222, config sifigan_config/ssc_postprocessing.yaml):

    f0_cvt, t = pw.harvest(x, fs, f0_floor, f0_ceil, frame_period=5)     :147   only len(f0_cvt) and t survive
    f0_       = lf0 of the decode CLI, np.interp'ed to len(f0_cvt), clamped at 0   :154-166
    sp        = pw.cheaptrick(x, f0_, t, fs)                             :167
    ap        = pw.d4c(x, f0_, t, fs)                                    :168
    mcep      = pysptk.sp2mc(sp, order=39, alpha=0.466)                  :169
    bap       = pw.code_aperiodicity(ap, fs)                             :171
    uv, cf0   = convert_continuos_f0(f0_)                                :174
    c         = [scaler["mcep"].transform(mcep), scaler["bap"].transform(bap)]   :185-199 (aux_feats mcep, bap)
    dfs       = np.repeat(dilated_factor(cf0, fs, df), us)               :201-210
    in_signal = SignalGenerator(fs, hop 120, sine_amp 0.1, noise_amp 0.003, ["sine"])(cf0)   :105-111,219-222

PINNED (in-tree numpy / scipy arithmetic, fixtures tests/golden/postproc_f0.npz made by importing the reference):
`convert_continuos_f0` (:51-72) and the `np.interp` length match (:159-164).

**PARITY UNPINNED** for everything else: pyworld (WORLD, M. Morise), pysptk (SPTK) and sifigan are third-party
packages absent from /root/reference and from the image, unpinned in the reference (README.md:22, setup.cfg), and
the reference holds no fixtures for them.  The functions below restate the PUBLISHED algorithms:
  * CheapTrick -- M. Morise, "CheapTrick, a spectral envelope estimator for high-quality speech synthesis", Speech
    Communication 67 (2015); "Error evaluation of an F0-adaptive spectral envelope estimator in robustness against
    the additive noise and F0 error", IEICE Trans. 2015 (the q1 = -0.15 compensation lifter), as implemented in WORLD
    (cheaptrick.cpp, common.cpp: DCCorrection, LinearSmoothing, interp1Q) with pyworld's defaults q1 = -0.15,
    f0_floor = 71 Hz => fft_size 1024 at 24 kHz;
  * D4C -- M. Morise, "D4C, a band-aperiodicity estimator for high-quality speech synthesis", Speech Communication 84
    (2016), as implemented in WORLD (d4c.cpp) incl. the "D4C Love Train" voicing check, threshold 0.85;
  * code_aperiodicity -- WORLD codec.cpp (3 kHz bands: 3 values at 24 kHz);
  * freqt / sp2mc / mc2sp -- SPTK's frequency transform recursion (Tokuda et al., mel-cepstral analysis) and pysptk's
    conversion helpers;
  * dilated_factor / SignalGenerator -- sifigan.utils.features (Yoneyama et al., Source-Filter HiFi-GAN, ICASSP 2023).
WORLD adds safeguard noise from its own `randn()` (1e-12 * randn per windowed sample, eps * |randn| per smoothed bin);
it cannot be reproduced and only matters for digital silence, so it is replaced here by its EXPECTED power
(deterministic): flat (2*hwl+1) * 1e-24 on the power spectrum, eps * sqrt(2/pi) on the smoothed spectrum.
"""
import numpy as np

K_PI = np.pi
SAFEGUARD = 1e-12            # world::kMySafeGuardMinimum
EPS = 2.2204460492503131e-16  # world::kEps
DEFAULT_F0 = 500.0            # world::kDefaultF0
FLOOR_F0_D4C = 47.0           # world::kFloorF0D4C
FREQ_INTERVAL = 3000.0        # world::kFrequencyInterval
UPPER_LIMIT = 15000.0         # world::kUpperLimit
NOISE_AFTER_SMOOTHING = EPS * np.sqrt(2.0 / np.pi)


def matlab_round(x):
    return int(x + 0.5) if x > 0 else int(x - 0.5)


# ------------------------------------------------------------------------------------------------ time axis
def harvest_frame_count(x_length, fs, frame_period=5.0):
    """GetSamplesForHarvest (harvest.cpp): the only thing the reference keeps of `pw.harvest` besides `t`."""
    return int(1000.0 * x_length / fs / frame_period) + 1


def harvest_time_axis(x_length, fs, frame_period=5.0):
    n = harvest_frame_count(x_length, fs, frame_period)
    return np.arange(n) * frame_period / 1000.0


def cheaptrick_fft_size(fs, f0_floor=71.0):
    return 2 ** (1 + int(np.log(3.0 * fs / f0_floor + 1) / np.log(2.0)))


# ------------------------------------------------------------------------------------------------ common.cpp
def interp1q(x0, shift, y, xi):
    """interp1Q: linear interpolation of y given on the grid x0 + shift * k, no bounds handling (as WORLD)."""
    pos = (np.asarray(xi, dtype=np.float64) - x0) / shift
    base = pos.astype(np.int64)  # truncation, as static_cast<int>
    frac = pos - base
    dy = np.diff(y, append=y[-1:])  # delta_y[x_length - 1] = 0
    return y[base] + dy[base] * frac


def dc_correction(inp, f0, fs, fft_size):
    out = np.array(inp, dtype=np.float64, copy=True)
    upper_limit = 2 + int(f0 * fft_size / fs)
    axis = np.arange(upper_limit) * fs / fft_size
    n_rep = upper_limit - 1
    rep = interp1q(f0 - axis[0], -float(fs) / fft_size, out[:upper_limit + 1], axis[:n_rep])
    out[:n_rep] += rep
    return out


def linear_smoothing(inp, width, fs, fft_size):
    half = fft_size // 2
    boundary = int(width * fft_size / fs) + 1
    mirror = np.empty(half + 2 * boundary + 1)
    mirror[:boundary] = inp[boundary:0:-1]
    mirror[boundary:half + boundary] = inp[:half]
    mirror[half + boundary:] = inp[half - np.arange(boundary + 1)]
    seg = np.cumsum(mirror * fs / fft_size)
    axis = np.arange(half + 1) / fft_size * fs - width / 2.0
    origin = -(boundary - 0.5) * fs / fft_size
    step = float(fs) / fft_size
    low = interp1q(origin, step, seg, axis)
    high = interp1q(origin, step, seg, axis + width)
    return (high - low) / width


def interp1(x, y, xi):
    """matlabfunctions.cpp interp1 for xi inside [x[0], x[-1]] (linear)."""
    return np.interp(xi, x, y)


# ------------------------------------------------------------------------------------------------ CheapTrick
def _safe_gather(x, origin, hwl):
    idx = np.clip(origin + np.arange(-hwl, hwl + 1), 0, len(x) - 1)
    return x[idx]


def _cheaptrick_windowed(x, fs, f0, position):
    hwl = matlab_round(1.5 * fs / f0)
    base = np.arange(-hwl, hwl + 1)
    origin = matlab_round(position * fs + 0.001)
    win = 0.5 * np.cos(K_PI * (base / 1.5 / fs) * f0) + 0.5
    win = win / np.sqrt(np.sum(win * win))
    wav = _safe_gather(x, origin, hwl) * win
    wav = wav - win * (np.sum(wav) / np.sum(win))
    return wav, hwl


def cheaptrick_frame(x, fs, f0, position, fft_size, q1=-0.15, want="sp"):
    """CheapTrickGeneralBody.  want = "sp" -> spectral envelope (fft_size/2+1); "cepstrum" -> the liftered cepstrum
    X[q] (WORLD's normalisation: the inverse transform of it is log(sp))."""
    half = fft_size // 2
    wav, hwl = _cheaptrick_windowed(x, fs, f0, position)
    spec = np.fft.rfft(wav, fft_size)
    power = spec.real ** 2 + spec.imag ** 2 + (2 * hwl + 1) * SAFEGUARD * SAFEGUARD
    power = dc_correction(power, f0, fs, fft_size)
    power = linear_smoothing(power, f0 * 2.0 / 3.0, fs, fft_size) + NOISE_AFTER_SMOOTHING
    quef = np.arange(1, half + 1) / fs
    smoothing = np.ones(half + 1)
    compensation = np.ones(half + 1)
    smoothing[1:] = np.sin(K_PI * f0 * quef) / (K_PI * f0 * quef)
    compensation[0] = (1.0 - 2.0 * q1) + 2.0 * q1
    compensation[1:] = (1.0 - 2.0 * q1) + 2.0 * q1 * np.cos(2.0 * K_PI * quef * f0)
    logp = np.log(power)
    sym = np.concatenate([logp, logp[half - 1:0:-1]])
    ceps = np.fft.rfft(sym).real
    lift = ceps * smoothing * compensation / fft_size
    if want == "cepstrum":
        return lift
    # c2r of a real "spectrum": X[0] + (-1)^k X[N/2] + 2 sum X[i] cos(2 pi i k / N)
    env = np.fft.irfft(lift, fft_size)[:half + 1] * fft_size
    return np.exp(env)


def cheaptrick(x, f0, t, fs, q1=-0.15, f0_floor=71.0, fft_size=None, want="sp"):
    """pyworld.cheaptrick(x, f0, temporal_positions, fs): (frames, fft_size/2+1) float64."""
    x = np.asarray(x, dtype=np.float64)
    if fft_size is None:
        fft_size = cheaptrick_fft_size(fs, f0_floor)
    floor = 3.0 * fs / (fft_size - 3.0)  # GetF0FloorForCheapTrick
    out = np.empty((len(f0), fft_size // 2 + 1))
    for i in range(len(f0)):
        cur = DEFAULT_F0 if f0[i] <= floor else float(f0[i])
        out[i] = cheaptrick_frame(x, fs, cur, t[i], fft_size, q1, want)
    return out


# ------------------------------------------------------------------------------------------------ D4C
def _d4c_windowed(x, fs, f0, position, window_type, ratio):
    hwl = matlab_round(ratio * fs / f0 / 2.0)
    base = np.arange(-hwl, hwl + 1)
    origin = matlab_round(position * fs + 0.001)
    pos = (2.0 * base / ratio) / fs
    if window_type == "hanning":
        win = 0.5 * np.cos(K_PI * pos * f0) + 0.5
    else:  # blackman
        win = 0.42 + 0.5 * np.cos(K_PI * pos * f0) + 0.08 * np.cos(K_PI * pos * f0 * 2)
    wav = _safe_gather(x, origin, hwl) * win
    wav = wav - win * (np.sum(wav) / np.sum(win))
    return wav


def nuttall_window(n):
    tmp = np.arange(n) / (n - 1.0)
    return (0.355768 - 0.487396 * np.cos(2.0 * K_PI * tmp) + 0.144232 * np.cos(4.0 * K_PI * tmp)
            - 0.012604 * np.cos(6.0 * K_PI * tmp))


def d4c_love_train(x, fs, f0, position):
    """D4CLoveTrainSub: share of the 100 Hz..4 kHz power in the 100 Hz..7.9 kHz power."""
    lowest_f0 = 40.0
    fft_size = 2 ** (1 + int(np.log(3.0 * fs / lowest_f0 + 1) / np.log(2.0)))
    b0 = int(np.ceil(100.0 * fft_size / fs))
    b1 = int(np.ceil(4000.0 * fft_size / fs))
    b2 = int(np.ceil(7900.0 * fft_size / fs))
    cur = max(f0, lowest_f0)
    wav = _d4c_windowed(x, fs, cur, position, "blackman", 3.0)
    spec = np.fft.rfft(wav, fft_size)
    p = spec.real ** 2 + spec.imag ** 2
    p[:b0 + 1] = 0.0
    c = np.cumsum(p)
    if c[b2] <= 0.0:
        return 0.0  # digital silence (WORLD: noise over noise); treated as unvoiced
    return c[b1] / c[b2]


def _d4c_centroid(x, fs, f0, position, fft_size):
    wav = _d4c_windowed(x, fs, f0, position, "blackman", 4.0)
    power = np.sum(wav * wav)
    if power <= 0.0:
        return None
    wav = wav / np.sqrt(power)
    buf = np.zeros(fft_size)
    buf[:len(wav)] = wav
    s1 = np.fft.rfft(buf)
    s2 = np.fft.rfft(buf * (np.arange(fft_size) + 1.0))
    return s2.real * s1.real + s1.imag * s2.imag


def d4c_frame(x, fs, f0, position, fft_size_d4c, n_bands, window):
    """D4CGeneralBody: coarse aperiodicity (dB) of the n_bands 3 kHz bands."""
    half = fft_size_d4c // 2
    c1 = _d4c_centroid(x, fs, f0, position - 0.25 / f0, fft_size_d4c)
    c2 = _d4c_centroid(x, fs, f0, position + 0.25 / f0, fft_size_d4c)
    if c1 is None or c2 is None:
        return None
    centroid = dc_correction(c1 + c2, f0, fs, fft_size_d4c)
    wav = _d4c_windowed(x, fs, f0, position, "hanning", 4.0)
    spec = np.fft.rfft(wav, fft_size_d4c)
    sm = dc_correction(spec.real ** 2 + spec.imag ** 2, f0, fs, fft_size_d4c)
    sm = linear_smoothing(sm, f0, fs, fft_size_d4c)
    gd = centroid / sm
    gd = linear_smoothing(gd, f0 / 2.0, fs, fft_size_d4c)
    gd = gd - linear_smoothing(gd, f0, fs, fft_size_d4c)
    wl = len(window)
    boundary = matlab_round(fft_size_d4c * 8.0 / wl)
    hw = wl // 2
    coarse = np.empty(n_bands)
    for i in range(n_bands):
        center = int(FREQ_INTERVAL * (i + 1) * fft_size_d4c / fs)
        seg = gd[center - hw:center + hw + 1] * window
        sp = np.fft.rfft(seg, fft_size_d4c)
        p = np.sort(sp.real ** 2 + sp.imag ** 2)
        c = np.cumsum(p)
        coarse[i] = 10 * np.log10(c[half - boundary - 1] / c[half])
    return np.minimum(0.0, coarse + (f0 - 100) / 50.0)


def d4c_band_aperiodicity(x, f0, t, fs, threshold=0.85):
    """The coarse band aperiodicities D4C computes, in dB: (frames, n_bands); unvoiced / rejected frames hold
    20*log10(1 - 1e-12), which is what code_aperiodicity returns for WORLD's initial value."""
    x = np.asarray(x, dtype=np.float64)
    fft_size_d4c = 2 ** (1 + int(np.log(4.0 * fs / FLOOR_F0_D4C + 1) / np.log(2.0)))
    n_bands = int(min(UPPER_LIMIT, fs / 2.0 - FREQ_INTERVAL) / FREQ_INTERVAL)
    wl = int(FREQ_INTERVAL * fft_size_d4c / fs) * 2 + 1
    window = nuttall_window(wl)
    out = np.full((len(f0), n_bands), 20 * np.log10(1.0 - SAFEGUARD))
    for i in range(len(f0)):
        if f0[i] == 0:
            continue
        if d4c_love_train(x, fs, float(f0[i]), t[i]) <= threshold:
            continue
        coarse = d4c_frame(x, fs, max(FLOOR_F0_D4C, float(f0[i])), t[i], fft_size_d4c, n_bands, window)
        if coarse is not None:
            out[i] = coarse
    return out


def d4c(x, f0, t, fs, threshold=0.85, fft_size=None):
    """pyworld.d4c: aperiodicity spectrogram (frames, fft_size/2+1), fft_size defaulting to CheapTrick's."""
    if fft_size is None:
        fft_size = cheaptrick_fft_size(fs)
    band = d4c_band_aperiodicity(x, f0, t, fs, threshold)
    n_bands = band.shape[1]
    coarse_axis = np.concatenate([np.arange(n_bands + 1) * FREQ_INTERVAL, [fs / 2.0]])
    axis = np.arange(fft_size // 2 + 1) * fs / fft_size
    ap = np.full((len(f0), fft_size // 2 + 1), 1.0 - SAFEGUARD)
    unvoiced = 20 * np.log10(1.0 - SAFEGUARD)
    for i in range(len(f0)):
        if np.all(band[i] == unvoiced):
            continue
        coarse = np.concatenate([[-60.0], band[i], [-SAFEGUARD]])
        ap[i] = 10.0 ** (interp1(coarse_axis, coarse, axis) / 20.0)
    return ap


def code_aperiodicity(ap, fs):
    """WORLD codec.cpp CodeAperiodicity: 20*log10(ap) read at 3, 6, 9 ... kHz."""
    n_bands = int(min(UPPER_LIMIT, fs / 2.0 - FREQ_INTERVAL) / FREQ_INTERVAL)
    fft_size = (ap.shape[1] - 1) * 2
    axis = np.arange(fft_size // 2 + 1) * fs / fft_size
    coarse_axis = FREQ_INTERVAL * (np.arange(n_bands) + 1.0)
    out = np.empty((ap.shape[0], n_bands))
    for i in range(ap.shape[0]):
        out[i] = interp1(axis, 20 * np.log10(ap[i]), coarse_axis)
    return out


# ------------------------------------------------------------------------------------------------ SPTK
def freqt(c, order, alpha):
    """SPTK freqt: frequency transformation of the cepstrum c[0..m1] to `order`+1 coefficients.  c may be (m1+1,)
    or (m1+1, n) -- n independent cepstra as columns (the recursion runs over m1 and the order, vectorised over n)."""
    c = np.asarray(c, dtype=np.float64)
    cols = c.reshape(c.shape[0], -1)
    m1 = cols.shape[0] - 1
    m2 = order
    b = 1 - alpha * alpha
    g = np.zeros((m2 + 1, cols.shape[1]))
    for i in range(-m1, 1):
        d = g.copy()
        g[0] = cols[-i] + alpha * d[0]
        if m2 >= 1:
            g[1] = b * d[0] + alpha * d[1]
        for j in range(2, m2 + 1):
            g[j] = d[j - 1] + alpha * (d[j] - g[j - 1])
    return g.reshape((m2 + 1,) + c.shape[1:])


def freqt_matrix(m1, order, alpha):
    """freqt is linear: (order+1, m1+1) matrix A with freqt(c) = A @ c."""
    return freqt(np.eye(m1 + 1), order, alpha)


def sp2mc(sp, order, alpha):
    """pysptk.sp2mc: log -> real cepstrum by irfft (c[0] halved) -> freqt.  Rows independent."""
    sp = np.atleast_2d(sp)
    c = np.fft.irfft(np.log(sp), axis=1)
    c[:, 0] /= 2.0
    return freqt(c.T, order, alpha).T


def mc2sp(mc, alpha, fftlen):
    """pysptk.mc2sp (the inverse direction, for round-trip tests)."""
    mc = np.atleast_2d(mc)
    c = freqt(mc.T, fftlen // 2, -alpha).T
    c[:, 0] *= 2.0
    sym = np.zeros((mc.shape[0], fftlen))
    sym[:, :c.shape[1]] = c
    sym[:, -1:-c.shape[1]:-1] = c[:, 1:]   # symc[-i] = c[i] (the two writes of c[N/2] coincide)
    return np.exp(np.fft.rfft(sym, axis=1).real)


# ------------------------------------------------------------------------------------------------ F0 features
def match_length(f0, n_out):
    """ssc_postprocessing.py:159-166."""
    f0 = np.asarray(f0)
    if len(f0) != n_out:
        x_orig = np.arange(len(f0))
        x_new = np.linspace(0, len(f0) - 1, n_out)
        f0 = np.maximum(np.interp(x_new, x_orig, f0.ravel()), 0)
    return f0.squeeze().astype(np.float64)


def convert_continuos_f0(f0):
    """ssc_postprocessing.py:51-72: (uv float32, continuous f0, ok)."""
    from scipy.interpolate import interp1d
    f0 = np.asarray(f0, dtype=np.float64)
    uv = np.float32(f0 != 0)
    if (f0 == 0).all():
        return uv, f0, False
    nz = np.nonzero(f0)[0]
    cont = f0.copy()
    cont[:nz[0]] = f0[nz[0]]
    cont[nz[-1]:] = f0[nz[-1]]
    nz = np.nonzero(cont)[0]
    return uv, interp1d(nz, cont[nz])(np.arange(0, cont.shape[0])), True


def dilated_factor(batch_f0, fs, dense_factor):
    """sifigan.utils.features.dilated_factor: fs / dense_factor / f0, unvoiced frames -> 1."""
    f0 = np.array(batch_f0, dtype=np.float64, copy=True)
    f0[f0 == 0] = fs / dense_factor
    out = np.ones(f0.shape) * fs / dense_factor / f0
    assert np.all(out > 0)
    return out


def signal_generator_sine(f0, sample_rate=24000, hop_size=120, sine_amp=0.1, noise_amp=0.003, noise=None):
    """sifigan.utils.features.SignalGenerator(signal_types=["sine"]).__call__(f0), f0 (B, 1, T) float32 tensor, as
    torch runs it on the CPU: nearest-neighbour upsampling, `radious = f0 / fs % 1`, cumsum (torch's CPU cumsum
    accumulates float32 inputs in double and rounds every output to float32), `sin(cumsum * 2 * pi)` in float32.
    `noise` (B, 1, T * hop) stands in for the reference's `torch.randn` draw; None -> no noise term."""
    import torch
    import torch.nn.functional as F
    f0 = torch.as_tensor(f0, dtype=torch.float32)
    B, _, T = f0.shape
    vuv = F.interpolate((f0 > 0) * torch.ones_like(f0), T * hop_size)
    rad = (F.interpolate(f0, T * hop_size) / sample_rate) % 1
    sine = vuv * torch.sin(torch.cumsum(rad, dim=2) * 2 * np.pi) * sine_amp
    if noise_amp > 0 and noise is not None:
        amp = vuv * noise_amp + (1.0 - vuv) * noise_amp / 3.0
        sine = sine + torch.as_tensor(noise, dtype=torch.float32) * amp
    return sine


# ------------------------------------------------------------------------------------------------ the stage
def pcm16_roundtrip(wave):
    """decode writes PCM_16 (libsndfile: lrint(x * 32767), clipped), post-processing reads it back as int16 / 32768."""
    q = np.clip(np.rint(np.asarray(wave, dtype=np.float64) * 32767.0), -32768, 32767)
    return q / 32768.0


def analyze(x, lf0, fs=24000, frame_period=5.0, mcep_dim=39, alpha=0.466,
            dense_factors=(0.5, 1, 4, 8), upsample_scales=(5, 4, 3, 2), mean=None, scale=None):
    """x float64 waveform, lf0 the decode CLI's contour -> dict(c (T, 43), cf0, uv, dfs [4], ok)."""
    t = harvest_time_axis(len(x), fs, frame_period)
    f0 = match_length(lf0, len(t))
    mcep = freqt_from_cepstrum(cheaptrick(x, f0, t, fs, want="cepstrum"), mcep_dim, alpha)
    bap = d4c_band_aperiodicity(x, f0, t, fs)
    uv, cf0, ok = convert_continuos_f0(f0)
    c = np.concatenate([mcep, bap], axis=1)
    if mean is not None:
        c = (c - mean) / scale
    dfs = []
    if ok:
        for df, us in zip(dense_factors, np.cumprod(upsample_scales)):
            dfs.append(np.repeat(dilated_factor(cf0[:, None], fs, df), us))
    return dict(c=c, mcep=mcep, bap=bap, f0=f0, cf0=cf0, uv=uv, dfs=dfs, ok=ok, t=t)


def freqt_from_cepstrum(lift, order, alpha):
    """sp2mc(exp(c2r(X))) without the exp / log / transform pair: irfft(log sp) returns X mirrored, so
    mc = freqt([X0 / 2, X1 .. X_{N/2}, X_{N/2-1} .. X1])."""
    lift = np.atleast_2d(lift)
    c = np.concatenate([lift, lift[:, -2:0:-1]], axis=1)
    c[:, 0] /= 2.0
    return freqt(c.T, order, alpha).T
