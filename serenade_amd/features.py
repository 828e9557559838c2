"""GPU feature front-end: the log-mel spectrogram and A-weighted loudness the reference computes with librosa on the
CPU before the hot path.

    logmelfilterbank(audio, sampling_rate, fft_size, hop_size, win_length, window, num_mels, fmin, fmax, eps,
                     log_base)                       serenade/bin/preprocess.py:140-203
    loudness_extract(audio, sampling_rate, hop_length)   serenade/bin/preprocess.py:126-137

Same names, arguments and return shapes ((#frames, num_mels) and (#frames,)); `audio` may also be a (B, n) batch of
equal-length signals, which adds a leading batch axis to the result.  Inputs are CUDA tensors (numpy arrays are
uploaded); everything runs in libserenade_hip.so: the STFT is a strided implicit GEMM (`srn_conv_gemm`, exact-fp32
MFMA) over the reflect-padded signal viewed as rows of 16 samples with window x DFT basis weights, followed by
`srn_logmel` / `srn_loudness`.  The constant tables (window, DFT basis, Slaney mel filterbank, A-weighting) are
built once on the host in float64.  There is no CPU path.

Parity: librosa is not vendored by the reference and not installed here, so this row is pinned to the restatement
in oracle/features_oracle.py only ("parity unpinned").
"""
import math

import numpy as np
import torch

from . import _lib, ops
from .models import _lru_get, _rup

__all__ = ["logmelfilterbank", "loudness_extract"]

_PLANS = {}


# ---------------------------------------------------------------------------------------------- constant tables
def _hann_padded(n_fft, win_length):
    n = np.arange(win_length, dtype=np.float64)
    w = np.zeros(n_fft)
    lp = (n_fft - win_length) // 2
    w[lp:lp + win_length] = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)  # periodic Hann, centred in the frame
    return w


def _dft_basis(n_fft, win_length, n_pad):
    """(n_pad, n_fft): rows [0, nb) = w cos, rows [nb, 2 nb) = -w sin (rfft sign), zero rows up to n_pad"""
    nb = 1 + n_fft // 2
    w = _hann_padded(n_fft, win_length)
    ang = 2.0 * np.pi * np.outer(np.arange(nb), np.arange(n_fft)) / n_fft
    basis = np.zeros((n_pad, n_fft))
    basis[:nb] = np.cos(ang) * w
    basis[nb:2 * nb] = -np.sin(ang) * w
    return basis.astype(np.float32)


def _slaney_mel(sr, n_fft, n_mels, fmin, fmax):
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    to_mel = lambda f: min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp
    mels = np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2)
    hz = np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    ramps = hz[:, None] - fft_f[None, :]
    d = np.diff(hz)
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]))
    return w * (2.0 / (hz[2:] - hz[:-2]))[:, None]  # Slaney area normalisation


def _a_weight_db(freqs, min_db=-80.0):
    f2 = np.asarray(freqs, dtype=np.float64) ** 2
    c = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2
    with np.errstate(divide="ignore"):
        w = 2.0 + 20.0 * (np.log10(c[0]) + 2 * np.log10(f2) - np.log10(f2 + c[0]) - np.log10(f2 + c[1])
                          - 0.5 * np.log10(f2 + c[2]) - 0.5 * np.log10(f2 + c[3]))
    return np.maximum(min_db, w)


# ---------------------------------------------------------------------------------------------- STFT plan
class _Stft:
    """reflect pad + strided implicit-GEMM STFT of (B, n) signals -> self.spec (B, frames, ld) = [re | im | 0]"""

    def __init__(self, dev, B, n, n_fft, hop, win_length, pad_mode="reflect"):
        if pad_mode not in ("reflect", "constant"):
            raise ValueError(f"pad_mode {pad_mode!r}: 'reflect' or 'constant'")
        c = math.gcd(math.gcd(n_fft, hop), 16)
        if c < 4:
            raise ValueError(f"fft_size {n_fft} and hop_size {hop} must share a factor of 4 (rows of the signal view)")
        self.B, self.n, self.nb = B, n, 1 + n_fft // 2
        self.frames = 1 + n // hop
        pad = n_fft // 2
        rows = -(-(n + 2 * pad) // c) + n_fft // c + 2  # the last frame's taps stay inside the buffer
        self.ld_sig = rows * c
        self.ld = _rup(2 * self.nb, 4)
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.audio, self.sig, self.spec = f(B, n), f(B, self.ld_sig), f(B, self.frames, self.ld)
        self.basis = torch.from_numpy(_dft_basis(n_fft, win_length, self.ld)).to(dev)
        self.ops = [ops.CallOp("srn_pad_signal", (self.audio, self.sig, B, n, pad, self.ld_sig, int(pad_mode == "constant")))]
        # A frame is a contraction over n_fft samples = taps of `cw` samples each.  The signal is viewed as rows that START
        # every c samples (row stride c, so any frame start is a row) but are cw = 32 samples WIDE (overlapping rows are
        # fine for a read-only operand): 32-channel taps take the fast contraction kernel, 16-wide ones the generic one
        # (loudness, n_fft 2048: 5.6 -> 0.9 ms per 8 x 1024 frames).
        cw = 32 if (c == 16 and n_fft % 32 == 0) else c
        step = cw // c
        taps_all = n_fft // cw
        for g0 in range(0, taps_all, _lib.SRN_MAX_TAPS):
            taps = [step * j for j in range(g0, min(g0 + _lib.SRN_MAX_TAPS, taps_all))]
            more = dict(res=self.spec, res_mode=ops.RES_ADD, res_bs=self.frames * self.ld, ld_res=self.ld) if g0 else {}
            self.ops.append(ops.ConvOp(in0=self.sig, w=(self.basis, g0 * cw), out=self.spec, n_batch=B, T_in=rows - step,
                                       T_out=self.frames, C_in=cw, N=self.ld, in0_bs=self.ld_sig, ld_in0=c, ldw=n_fft,
                                       out_bs=self.frames * self.ld, ld_out=self.ld, taps=taps, in_stride=hop // c,
                                       precision=_lib.PREC_FP32, **more))

    def load(self, audio):
        self.audio.copy_(audio.reshape(self.B, self.n), non_blocking=True)


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: the HIP path needs CUDA (ROCm) tensors; there is no CPU fallback")


def _as_batch(audio):
    if not isinstance(audio, torch.Tensor):  # numpy input, as the reference's callers pass: uploaded
        if not torch.cuda.is_available():
            raise RuntimeError("serenade_amd.features needs a CUDA (ROCm) device; there is no CPU fallback")
        audio = torch.as_tensor(np.asarray(audio, dtype=np.float32)).cuda()
    _require_cuda(audio, "serenade_amd.features")
    a = audio.detach().to(torch.float32)
    return (a.unsqueeze(0), True) if a.dim() == 1 else (a, False)


def logmelfilterbank(audio, sampling_rate, fft_size=1024, hop_size=256, win_length=None, window="hann", num_mels=80,
                     fmin=None, fmax=None, eps=1e-10, log_base=10.0):
    """preprocess.py:140-203: (#frames, num_mels) log-mel spectrogram (leading batch axis for (B, n) input)."""
    if window != "hann":
        raise ValueError("only the recipe's Hann window is implemented")
    if log_base not in (None, 10.0, 2.0):
        raise ValueError(f"{log_base} is not supported.")
    a, single = _as_batch(audio)
    B, n = a.shape
    win_length = fft_size if win_length is None else win_length
    fmin = 0 if fmin is None else fmin
    fmax = sampling_rate / 2 if fmax is None else fmax
    key = ("mel", str(a.device), B, n, sampling_rate, fft_size, hop_size, win_length, num_mels, fmin, fmax, eps, log_base)

    def make():
        st = _Stft(a.device, B, n, fft_size, hop_size, win_length)
        mel_t = torch.from_numpy(_slaney_mel(sampling_rate, fft_size, num_mels, fmin, fmax).T.astype(np.float32).copy())
        mel_t = mel_t.to(a.device).contiguous()
        out = torch.zeros(B, st.frames, num_mels, device=a.device, dtype=torch.float32)
        mode = 0 if log_base is None else int(log_base)
        op = ops.CallOp("srn_logmel", (st.spec, mel_t, out, B * st.frames, st.nb, st.ld, num_mels, float(eps), mode))
        return st, mel_t, out, st.ops + [op]

    st, _, out, ol = _lru_get(_PLANS, key, 8, make)
    st.load(a)
    for op in ol:
        op()
    res = out.clone()
    return res[0] if single else res


def loudness_extract(audio, sampling_rate, hop_length, pad_mode="constant"):
    """preprocess.py:126-137: (#frames,) log mean A-weighted amplitude (librosa defaults: n_fft 2048, Hann,
    power_to_db top_db 80 relative to the utterance's loudest bin).

    pad_mode: the reference calls `librosa.stft(audio, hop_length=hop_length)` without a pad_mode, so the edge frames
    depend on the installed librosa: "constant" (zeros) since librosa 0.10 -- the default here --, "reflect" before
    (setup.cfg only asks for librosa >= 0.8.0).  Use the mode the `lft` features of a checkpoint were extracted with;
    about n_fft / 2 / hop frames at each end of an utterance differ between the two."""
    n_fft = 2048
    a, single = _as_batch(audio)
    B, n = a.shape
    key = ("loud", str(a.device), B, n, sampling_rate, hop_length, pad_mode)

    def make():
        st = _Stft(a.device, B, n, n_fft, hop_length, n_fft, pad_mode)
        aw = torch.from_numpy(_a_weight_db(np.linspace(0.0, sampling_rate / 2.0, st.nb)).astype(np.float32)).to(a.device)
        ws = torch.zeros(B, device=a.device, dtype=torch.int32)
        out = torch.zeros(B, st.frames, device=a.device, dtype=torch.float32)
        op = ops.CallOp("srn_loudness", (st.spec, aw, ws, out, B, st.frames, st.nb, st.ld, 1e-10, 80.0, 1e-5))
        return st, (aw, ws), out, st.ops + [op]

    st, _, out, ol = _lru_get(_PLANS, key, 8, make)
    st.load(a)
    for op in ol:
        op()
    res = out.clone()
    return res[0] if single else res
