"""Pitch bookkeeping of the decode loop (host numpy; nothing here is on the GPU path).

Behavioural mirror of the helpers the reference keeps inside its CLI (serenade/bin/ssc_decode.py:32-154).  The
public names are the reference's because its post-processing stage imports them by name; the implementation is
organised around one pair of pitch-scale maps (Hz <-> cents above a C4 anchor) and one decision function for the
style-transfer transposition.  Arithmetic that ends up in output files (`lf0` datasets) keeps the reference's
operation order so the written values are bit-identical.
"""
import numpy as np

#: C4 = three semitones above A3 = 440 Hz * 2^(3/12 - 1); anchored at 4800 cents so that voiced frames stay > 0
C4_HZ = 440 * 2 ** (3 / 12 - 1)
C4_CENT = 4800
_LN2 = np.log(2)

#: asymmetric stretch of the transposition: a target above the source is over-shot, one below is under-shot
_UP_GAIN = 1.4
_DOWN_GAIN = 5 / 7


def _voiced_log(f0):
    """natural log of the voiced (non-zero) frames of one contour"""
    f0 = np.asarray(f0)
    return np.log(f0[np.nonzero(f0)])


def _cents_of(hz):
    """cents above the C4 anchor of strictly positive frequencies (no unvoiced handling)"""
    return 1200 * np.log(hz / C4_HZ) / _LN2 + C4_CENT


def _hz_of(cent):
    return np.exp((cent - C4_CENT) * _LN2 / 1200) * C4_HZ


def _map_positive(values, fn):
    """copy of `values` with `fn` applied where values > 0 (zeros mark unvoiced frames and stay zero)"""
    out = values.copy()
    sel = np.where(values > 0)[0]
    out[sel] = fn(values[sel])
    return out


def hz_to_cent_based_c4(hz):
    return _map_positive(hz, _cents_of)


def cent_to_hz_based_c4(cent):
    return _map_positive(cent, _hz_of)


class F0Statistics(object):
    """Gaussian statistics of log-F0 over voiced frames, and the mean/variance-matching conversion."""

    def estimate(self, f0list):
        pooled = np.concatenate([_voiced_log(f0) for f0 in f0list]) if len(f0list) > 1 else _voiced_log(f0list[0])
        return np.array([np.mean(pooled), np.std(pooled)])

    def convert(self, f0, orgf0stats, tarf0stats):
        (mu_s, sd_s), (mu_t, sd_t) = orgf0stats, tarf0stats
        out = np.zeros(len(f0))
        voiced = f0 > 0
        out[voiced] = np.exp((sd_t / sd_s) * (np.log(f0[voiced]) - mu_s) + mu_t)
        return out


def transposition_cents(src_f0, trg_f0):
    """Whole-semitone transposition (in cents) that moves the source's mean log-pitch towards the target's:
    the raw distance is stretched by 1.4 upwards / 5/7 downwards and rounded to the nearest 100 cents."""
    stats = F0Statistics()
    mean_cent = [_cents_of(np.exp(stats.estimate([c])[0])) for c in (src_f0, trg_f0)]
    gap = mean_cent[1] - mean_cent[0]
    gain = _UP_GAIN if gap >= 0 else _DOWN_GAIN
    return round(gap * gain / 100) * 100


def linear_midi_shift(sm, tm):
    """Transpose the source contour `sm` (Hz) by `transposition_cents(sm, tm)`.  Like the reference
    (ssc_decode.py:130-154) the voiced frames of `sm` are overwritten IN PLACE and `sm` is returned; frames whose
    shifted pitch would fall below the anchor's zero are clamped there."""
    shift = transposition_cents(sm, tm)
    voiced = sm > 0
    sm[voiced] = hz_to_cent_based_c4(sm[voiced])
    sm[voiced] = np.maximum(0, sm[voiced] + shift)
    sm[voiced] = cent_to_hz_based_c4(sm[voiced])
    return sm
