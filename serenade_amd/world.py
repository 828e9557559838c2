"""Analysis front-end between HiFi-GAN and SiFiGAN on the MI355X -- SURVEY.md section 8 row f1.

The reference's stage 9 (serenade/bin/ssc_postprocessing.py:142-222) re-analyses every converted waveform on the CPU
with three third-party C / Python packages before it can call the SiFiGAN generator:

    f0_cvt, t = pw.harvest(x, fs, ...)               :147   only len(f0_cvt) and t are used -> `harvest_time_axis`
    f0_ = np.interp(...lf0 of the decode CLI...)     :154-166                                -> `match_length`
    sp = pw.cheaptrick(x, f0_, t, fs)                :167                                    -> `cheaptrick`
    ap = pw.d4c(x, f0_, t, fs); bap = pw.code_aperiodicity(ap, fs)   :168,171               -> `d4c_band_aperiodicity`
    mcep = pysptk.sp2mc(sp, order=39, alpha=0.466)   :169                                    -> `sp2mc`
    uv, cf0, ok = convert_continuos_f0(f0_)          :51-72,174                              -> `convert_continuos_f0`
    dfs = np.repeat(dilated_factor(cf0, fs, df), us) :201-210                                -> `Analyzer` (excitation)
    in_signal = SignalGenerator(...)(cf0)            :105-111,219-222                        -> `Analyzer` (excitation)

(`mcap = sp2mc(ap)` :170 is computed and never used with aux_feats = [mcep, bap]; Harvest's F0 track is discarded.)
Here the same functions take and return CUDA tensors and run in libserenade_hip.so (csrc/world.hip, float64, one
workgroup per 5 ms frame); `Analyzer` chains them so that waveform -> (c, dfs, in_signal) never leaves the GPU.  Host
code only builds constant tables (FFT twiddles, the Nuttall window, the sp2mc matrix) in float64 numpy.
There is no CPU path.

Parity: pyworld / pysptk / sifigan are not under /root/reference and not installed: everything except
`convert_continuos_f0` and `match_length` (pinned by tests/golden/postproc_f0.npz) is **parity unpinned** and is checked
against the restatement of the published algorithms in oracle/world_oracle.py (tests only).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import SrnExcitationParams, SrnWorldParams, check
from .ops import _stream

__all__ = ["ALPHA", "harvest_frame_count", "harvest_time_axis", "cheaptrick", "cheaptrick_fft_size",
           "d4c_band_aperiodicity", "sp2mc", "match_length", "convert_continuos_f0", "Analyzer"]

# all-pass constants of the mel-cepstral warp per sampling rate (ssc_postprocessing.py:39-48; public SPTK values)
ALPHA = {8000: 0.312, 12000: 0.369, 16000: 0.410, 22050: 0.455, 24000: 0.466, 32000: 0.504, 44100: 0.544,
         48000: 0.554}

_F64 = torch.float64
_UNVOICED_DB = float(20 * np.log10(1.0 - 1e-12))


# ------------------------------------------------------------------------------------------------ sizes (host)
def harvest_frame_count(x_length, fs, frame_period=5.0):
    """number of frames `pw.harvest` returns (WORLD GetSamplesForHarvest); the reference keeps nothing else of it."""
    return int(1000.0 * x_length / fs / frame_period) + 1


def harvest_time_axis(x_length, fs, frame_period=5.0):
    """temporal positions `t` of `pw.harvest` in seconds (float64 numpy)."""
    return np.arange(harvest_frame_count(x_length, fs, frame_period)) * frame_period / 1000.0


def cheaptrick_fft_size(fs, f0_floor=71.0):
    """WORLD GetFFTSizeForCheapTrick: 1024 at 24 kHz with pyworld's default floor."""
    return 2 ** (1 + int(math.log(3.0 * fs / f0_floor + 1) / math.log(2.0)))


def _d4c_fft_size(fs):
    return 2 ** (1 + int(math.log(4.0 * fs / 47.0 + 1) / math.log(2.0)))


def _n_bands(fs):
    return int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)


# ------------------------------------------------------------------------------------------------ constant tables
_TABLES = {}


def _table(dev, key, make):
    k = (dev.index, key)
    if k not in _TABLES:
        _TABLES[k] = torch.from_numpy(np.ascontiguousarray(make(), dtype=np.float64)).to(dev)
    return _TABLES[k]


def _twiddles(dev, n):
    def make():
        ang = 2.0 * np.pi * np.arange(n // 2) / n
        return np.stack([np.cos(ang), -np.sin(ang)], axis=1)
    return _table(dev, ("tw", n), make)


def _nuttall(dev, n):
    def make():
        tmp = np.arange(n) / (n - 1.0)
        return (0.355768 - 0.487396 * np.cos(2.0 * np.pi * tmp) + 0.144232 * np.cos(4.0 * np.pi * tmp)
                - 0.012604 * np.cos(6.0 * np.pi * tmp))
    return _table(dev, ("nuttall", n), make)


def _freqt_matrix(m1, order, alpha):
    """SPTK's frequency-transform recursion applied to the identity: (order+1, m1+1) with freqt(c) = A @ c."""
    beta = 1.0 - alpha * alpha
    g = np.zeros((order + 1, m1 + 1))
    eye = np.eye(m1 + 1)
    for i in range(m1, -1, -1):
        prev = g.copy()
        g[0] = eye[i] + alpha * prev[0]
        if order >= 1:
            g[1] = beta * prev[0] + alpha * prev[1]
        for j in range(2, order + 1):
            g[j] = prev[j - 1] + alpha * (prev[j] - g[j - 1])
    return g


def _sp2mc_matrix(dev, n_bins, order, alpha, from_cepstrum):
    """(n_bins, order+1) float64, transposed for the kernel.  from_cepstrum False: mc = M @ log(sp) (irfft, c0 / 2 and
    freqt are linear); True: mc = M @ X for the liftered cepstrum X CheapTrick leaves (irfft(log sp) is X mirrored)."""
    def make():
        fftlen = 2 * (n_bins - 1)
        a = _freqt_matrix(fftlen - 1, order, alpha)
        a[:, 0] *= 0.5
        if from_cepstrum:
            m = a[:, :n_bins].copy()
            m[:, 1:n_bins - 1] += a[:, :n_bins - 1:-1]
            return m.T
        return (a @ np.fft.irfft(np.eye(n_bins), axis=1).T).T
    return _table(dev, ("sp2mc", n_bins, order, alpha, from_cepstrum), make)


# ------------------------------------------------------------------------------------------------ argument plumbing
def _require_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError(f"{what}: expected a CUDA tensor (this module has no CPU path)")


def _as_f64(v, dev):
    if torch.is_tensor(v):
        return v.to(device=dev, dtype=_F64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev)


def _i32(values, dev):
    return torch.tensor([int(v) for v in values], dtype=torch.int32, device=dev)


def _batch_view(x, f0, t):
    """pyworld passes one utterance as 1-D arrays; a leading batch axis is accepted as well (equal lengths)."""
    _require_cuda(x, "x")
    dev = x.device
    single = x.dim() == 1
    x = _as_f64(x, dev)
    f0, t = _as_f64(f0, dev), _as_f64(t, dev)
    if single:
        x, f0, t = x[None], f0[None], t[None]
    if t.dim() == 1:
        t = t[None].expand_as(f0).contiguous()
    assert x.dim() == 2 and f0.dim() == 2 and f0.shape == t.shape and f0.size(0) == x.size(0)
    return x, f0, t, single


def _world_params(x, x_len, f0, t, n_frames, fs, fft_size):
    B, F = f0.shape
    p = SrnWorldParams()
    p.n_batch, p.max_frames, p.fs, p.fft_size = B, F, int(fs), int(fft_size)
    p.x, p.x_bs, p.x_len = x.data_ptr(), x.stride(0), x_len.data_ptr()
    p.f0, p.t, p.f_bs, p.n_frames = f0.data_ptr(), t.data_ptr(), f0.stride(0), n_frames.data_ptr()
    p.twiddle = _twiddles(x.device, fft_size).data_ptr()
    p.unvoiced_db = _UNVOICED_DB
    return p


def _check_f0(f0, fs):
    # WORLD itself has no bound; windows of 3-4 periods and the DC-correction span need F0 well inside the spectrum.
    # Negative values would come out of `dilated_factor` as negative dilations (the reference asserts there).
    if not f0.numel():
        raise ValueError("empty F0 contour")
    lo, top = (float(v) for v in torch.stack([f0.min(), f0.max()]).tolist())
    if not top < fs / 4.0:
        raise ValueError(f"F0 contour reaches {top} Hz; analysis needs F0 < fs / 4 = {fs / 4.0} Hz")
    if not lo >= 0.0:
        raise ValueError(f"F0 contour holds negative or non-finite values (min {lo})")


def _cheaptrick_raw(x, x_len, f0, t, n_frames, fs, q1, f0_floor, fft_size, want_sp, want_ceps):
    dev = x.device
    B, F = f0.shape
    nb = fft_size // 2 + 1
    p = _world_params(x, x_len, f0, t, n_frames, fs, fft_size)
    p.q1 = float(q1)
    p.f0_floor = 3.0 * fs / (fft_size - 3.0)  # GetF0FloorForCheapTrick: the lowest F0 the window still fits
    del f0_floor  # pyworld's argument only sizes the transform
    sp = torch.empty(B, F, nb, dtype=_F64, device=dev) if want_sp else None
    ceps = torch.empty(B, F, nb, dtype=_F64, device=dev) if want_ceps else None
    if sp is not None:
        p.out0, p.out0_bs, p.ld_out0 = sp.data_ptr(), sp.stride(0), nb
    if ceps is not None:
        p.out1, p.out1_bs, p.ld_out1 = ceps.data_ptr(), ceps.stride(0), nb
    check(_lib.lib().srn_world_cheaptrick(ctypes.byref(p), _stream()), "srn_world_cheaptrick")
    return sp, ceps


def cheaptrick(x, f0, temporal_positions, fs, q1=-0.15, f0_floor=71.0, fft_size=None):
    """`pyworld.cheaptrick`: spectral envelope (frames, fft_size/2+1) float64 -- (B, frames, .) for batched input."""
    x, f0, t, single = _batch_view(x, f0, temporal_positions)
    _check_f0(f0, fs)
    if fft_size is None:
        fft_size = cheaptrick_fft_size(fs, f0_floor)
    dev = x.device
    x_len = _i32([x.size(1)] * x.size(0), dev)
    n_frames = _i32([f0.size(1)] * x.size(0), dev)
    sp, _ = _cheaptrick_raw(x, x_len, f0, t, n_frames, fs, q1, f0_floor, fft_size, True, False)
    return sp[0] if single else sp


def _d4c_raw(x, x_len, f0, t, n_frames, fs, threshold):
    dev = x.device
    B, F = f0.shape
    N = _d4c_fft_size(fs)
    nb = _n_bands(fs)
    wl = int(3000.0 * N / fs) * 2 + 1
    p = _world_params(x, x_len, f0, t, n_frames, fs, N)
    p.threshold = float(threshold)
    win = _nuttall(dev, wl)
    p.band_window, p.band_window_len, p.n_bands = win.data_ptr(), wl, nb
    bap = torch.empty(B, F, nb, dtype=_F64, device=dev)
    p.out0, p.out0_bs, p.ld_out0 = bap.data_ptr(), bap.stride(0), nb
    check(_lib.lib().srn_world_d4c(ctypes.byref(p), _stream()), "srn_world_d4c")
    return bap


def d4c_band_aperiodicity(x, f0, temporal_positions, fs, threshold=0.85):
    """`pyworld.code_aperiodicity(pyworld.d4c(x, f0, t, fs), fs)`: (frames, n_bands) float64 in dB.  The full
    aperiodicity spectrogram is never materialised: D4C estimates one value per 3 kHz band, pyworld interpolates them
    onto the FFT grid and code_aperiodicity reads the grid back at the band centres."""
    x, f0, t, single = _batch_view(x, f0, temporal_positions)
    _check_f0(f0, fs)
    dev = x.device
    x_len = _i32([x.size(1)] * x.size(0), dev)
    n_frames = _i32([f0.size(1)] * x.size(0), dev)
    bap = _d4c_raw(x, x_len, f0, t, n_frames, fs, threshold)
    return bap[0] if single else bap


def _project(inp, mat_t, take_log):
    rows, K = inp.numel() // inp.size(-1), inp.size(-1)
    n_out = mat_t.size(1)
    out = torch.empty(*inp.shape[:-1], n_out, dtype=_F64, device=inp.device)
    check(_lib.lib().srn_world_project(inp.data_ptr(), rows, K, K, mat_t.data_ptr(), n_out, int(take_log),
                                       out.data_ptr(), n_out, _stream()), "srn_world_project")
    return out


def sp2mc(powerspec, order, alpha):
    """`pysptk.sp2mc`: mel-cepstrum (…, order+1) float64 of a power spectral envelope (…, fftlen/2+1)."""
    _require_cuda(powerspec, "powerspec")
    sp = powerspec.to(_F64).contiguous()
    return _project(sp, _sp2mc_matrix(sp.device, sp.size(-1), int(order), float(alpha), False), True)


def match_length(f0_list, n_out, device):
    """the np.interp length match of ssc_postprocessing.py:159-166 for a list of contours -> (B, max n_out) float64."""
    B = len(f0_list)
    n_in = [int(np.asarray(f).size) for f in f0_list]
    buf = np.zeros((B, max(n_in)))
    for i, f in enumerate(f0_list):
        buf[i, :n_in[i]] = np.asarray(f, dtype=np.float64).ravel()
    src = torch.from_numpy(buf).to(device)
    out = torch.zeros(B, max(n_out), dtype=_F64, device=device)
    n_in_d, n_out_d = _i32(n_in, device), _i32(n_out, device)  # named: the pointers must outlive the launch call
    check(_lib.lib().srn_f0_match_length(src.data_ptr(), src.stride(0), n_in_d.data_ptr(), out.data_ptr(),
                                         out.stride(0), n_out_d.data_ptr(), B, max(n_out), _stream()),
          "srn_f0_match_length")
    return out


def _cont_f0_raw(f0, n_frames):
    B, F = f0.shape
    cf0 = torch.zeros_like(f0)
    uv = torch.zeros(B, F, dtype=torch.float32, device=f0.device)
    ok = torch.zeros(B, dtype=torch.int32, device=f0.device)
    check(_lib.lib().srn_cont_f0(f0.data_ptr(), f0.stride(0), n_frames.data_ptr(), cf0.data_ptr(), uv.data_ptr(),
                                 ok.data_ptr(), B, _stream()), "srn_cont_f0")
    return uv, cf0, ok


def convert_continuos_f0(f0):
    """ssc_postprocessing.py:51-72 for one contour (CUDA float64 tensor): (uv float32, cont_f0 float64, ok bool)."""
    _require_cuda(f0, "f0")
    f = f0.to(_F64).reshape(1, -1).contiguous()
    uv, cf0, ok = _cont_f0_raw(f, _i32([f.size(1)], f.device))
    return uv[0], cf0[0], bool(ok.item())


class Analyzer:
    """waveform + decode-CLI F0 contour -> SiFiGAN's inputs, all on the GPU.

    Keyword names follow sifigan_config/ssc_postprocessing.yaml (:25-37) and generator/sifigan.yaml (upsample_scales).
    `scaler` maps "mcep" / "bap" to objects with `mean_` / `scale_` (the StandardScalers of `stats`), or None."""

    def __init__(self, sample_rate=24000, frame_period=5, mcep_dim=39, dense_factors=(0.5, 1, 4, 8),
                 upsample_scales=(5, 4, 3, 2), df_f0_type="cf0", sine_amp=0.1, noise_amp=0.003, sine_f0_type="cf0",
                 signal_types=("sine",), aux_feats=("mcep", "bap"), scaler=None, pcm16=True):
        if list(signal_types) != ["sine"] or list(aux_feats) != ["mcep", "bap"]:
            raise NotImplementedError("built for the reference's configuration: signal_types [sine], aux_feats [mcep, bap]")
        if len(dense_factors) > 4:
            raise ValueError("at most 4 dilated-factor tracks")
        self.fs, self.frame_period, self.order = int(sample_rate), float(frame_period), int(mcep_dim)
        self.alpha = ALPHA[self.fs]
        self.hop = int(self.fs * frame_period * 0.001)
        self.dense_factors = [float(d) for d in dense_factors]
        self.upsample = [int(u) for u in np.cumprod(upsample_scales)][:len(dense_factors)]
        if self.upsample and self.upsample[-1] > self.hop:
            raise ValueError("upsample_scales exceed the analysis hop")
        self.df_f0_type, self.sine_f0_type = df_f0_type, sine_f0_type
        self.sine_amp, self.noise_amp = float(sine_amp), float(noise_amp)
        self.pcm16 = bool(pcm16)
        self._stats = None
        if scaler is not None:
            mean = np.concatenate([np.asarray(scaler[k].mean_, dtype=np.float64).ravel() for k in aux_feats])
            scale = np.concatenate([np.asarray(scaler[k].scale_, dtype=np.float64).ravel() for k in aux_feats])
            self._stats = (mean, scale)
        self.fft_size = cheaptrick_fft_size(self.fs)

    def features(self, wave, lengths, f0_list):
        """wave (B, N) float32 CUDA (converted audio), lengths (B,) valid samples, f0_list: B contours (the `lf0`
        datasets of the decode CLI).  Returns dict(c (B, 43, F) float32, mcep, bap (float64), f0, cf0 (B, F) float64,
        uv, ok (B,) int32, n_frames list)."""
        _require_cuda(wave, "wave")
        dev = wave.device
        wave = wave.to(torch.float32).reshape(len(f0_list), -1).contiguous()
        B = wave.size(0)
        lengths = [int(v) for v in lengths]
        if len(lengths) != B or min(lengths) < 1 or max(lengths) > wave.size(1):
            raise ValueError(f"lengths {lengths} do not describe a (B = {B}, N = {wave.size(1)}) batch of waveforms")
        if any(np.asarray(f).size < 1 for f in f0_list):
            raise ValueError("every item needs an F0 contour of at least one frame")
        n_frames = [harvest_frame_count(n, self.fs, self.frame_period) for n in lengths]
        F = max(n_frames)
        x = torch.empty(wave.shape, dtype=_F64, device=dev)
        check(_lib.lib().srn_wave_to_f64(wave.data_ptr(), x.data_ptr(), wave.numel(), int(self.pcm16), _stream()),
              "srn_wave_to_f64")
        f0 = match_length(f0_list, n_frames, dev)
        _check_f0(f0, self.fs)
        t = torch.from_numpy(np.arange(F) * self.frame_period / 1000.0).to(dev)[None].expand(B, F).contiguous()
        x_len, nf = _i32(lengths, dev), _i32(n_frames, dev)
        _, ceps = _cheaptrick_raw(x, x_len, f0, t, nf, self.fs, -0.15, 71.0, self.fft_size, False, True)
        mcep = _project(ceps, _sp2mc_matrix(dev, ceps.size(-1), self.order, self.alpha, True), False)
        bap = _d4c_raw(x, x_len, f0, t, nf, self.fs, 0.85)
        nc = mcep.size(-1) + bap.size(-1)
        c = torch.zeros(B, F, nc, dtype=torch.float32, device=dev)
        mean = scale = None
        if self._stats is not None:
            mean, scale = (_table(dev, ("stats", i, self._stats[i].tobytes()), lambda i=i: self._stats[i]) for i in (0, 1))
        check(_lib.lib().srn_world_pack_features(mcep.data_ptr(), mcep.size(-1), bap.data_ptr(), bap.size(-1),
                                                 mean.data_ptr() if mean is not None else None,
                                                 scale.data_ptr() if scale is not None else None,
                                                 c.data_ptr(), B * F, nc, _stream()), "srn_world_pack_features")
        uv, cf0, ok = _cont_f0_raw(f0, nf)
        return dict(c=c.transpose(1, 2), mcep=mcep, bap=bap, f0=f0, cf0=cf0, uv=uv, ok=ok, n_frames=n_frames, nf=nf)

    def excitation(self, feats, noise=None, generator=None):
        """SiFiGAN's source inputs from `features()`: in_signal (B, 1, F * hop) float32 and the dilated-factor tracks
        [(B, 1, F * us_i)].  noise: (B, 1, F * hop) standard-normal draws (the reference's torch.randn) -- drawn here
        with `generator` when None and noise_amp > 0."""
        f0, cf0, nf = feats["f0"], feats["cf0"], feats["nf"]
        dev = f0.device
        B, F = f0.shape
        n = F * self.hop
        if noise is None and self.noise_amp > 0:
            noise = torch.randn((B, 1, n), device=dev, generator=generator)
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).reshape(B, n).contiguous()
        sine = torch.zeros(B, 1, n, dtype=torch.float32, device=dev)
        dfs = [torch.zeros(B, 1, F * u, dtype=torch.float32, device=dev) for u in self.upsample]
        phase = torch.empty(B, F, dtype=_F64, device=dev)
        p = SrnExcitationParams()
        p.n_batch, p.max_frames, p.fs, p.hop = B, F, self.fs, self.hop
        p.f0 = (cf0 if self.sine_f0_type == "cf0" else f0).data_ptr()
        p.df_f0 = (cf0 if self.df_f0_type == "cf0" else f0).data_ptr()
        p.f_bs, p.n_frames, p.phase_ws = f0.stride(0), nf.data_ptr(), phase.data_ptr()
        p.noise = noise.data_ptr() if noise is not None else None
        p.sine, p.sine_amp, p.noise_amp, p.n_df = sine.data_ptr(), self.sine_amp, self.noise_amp, len(dfs)
        for i, d in enumerate(dfs):
            p.dfs[i], p.df_upsample[i], p.dense_factors[i] = d.data_ptr(), self.upsample[i], self.dense_factors[i]
        check(_lib.lib().srn_sifigan_excitation(ctypes.byref(p), _stream()), "srn_sifigan_excitation")
        return sine, dfs

    def __call__(self, wave, lengths, f0_list, noise=None, generator=None):
        feats = self.features(wave, lengths, f0_list)
        in_signal, dfs = self.excitation(feats, noise, generator)
        return in_signal, feats["c"], dfs, feats
