"""F0 helpers of the decode loop (host-side numpy), same behaviour as serenade/bin/ssc_decode.py:32-154:
log-F0 statistics, Hz <-> cent (C4-based) and the semitone-rounded `linear_midi_shift`."""
import numpy as np

C4_HZ = 440 * 2 ** (3 / 12 - 1)
C4_CENT = 4800


class F0Statistics(object):
    """mean / std of log-F0 over voiced frames and the classic linear log-F0 conversion."""

    def estimate(self, f0list):
        logs = [np.log(f0[np.nonzero(f0)]) for f0 in f0list]
        f0s = np.concatenate(logs) if len(logs) > 1 else logs[0]
        return np.array([np.mean(f0s), np.std(f0s)])

    def convert(self, f0, orgf0stats, tarf0stats):
        cvf0 = np.zeros(len(f0))
        v = f0 > 0
        cvf0[v] = np.exp((tarf0stats[1] / orgf0stats[1]) * (np.log(f0[v]) - orgf0stats[0]) + tarf0stats[0])
        return cvf0


def hz_to_cent_based_c4(hz):
    out = hz.copy()
    nz = np.where(hz > 0)[0]
    out[nz] = 1200 * np.log(hz[nz] / C4_HZ) / np.log(2) + C4_CENT
    return out


def cent_to_hz_based_c4(cent):
    out = cent.copy()
    nz = np.where(cent > 0)[0]
    out[nz] = np.exp((cent[nz] - C4_CENT) * np.log(2) / 1200) * C4_HZ
    return out


def linear_midi_shift(sm, tm):
    """Shift the source F0 contour `sm` (Hz, modified IN PLACE like the reference, and returned) towards the
    mean pitch of `tm` by a whole number of semitones: upward shifts are scaled by 1.4, downward by 5/7, then
    rounded to 100 cents (ssc_decode.py:130-154)."""
    stats = F0Statistics()
    idx_s = sm > 0
    src = stats.estimate([sm])
    trg = stats.estimate([tm])
    src_cent = 1200 * np.log(np.exp(src[0]) / C4_HZ) / np.log(2) + C4_CENT
    tgt_cent = 1200 * np.log(np.exp(trg[0]) / C4_HZ) / np.log(2) + C4_CENT
    d = tgt_cent - src_cent
    shift = round(d * 1.4 / 100) * 100 if d >= 0 else round(d * (5 / 7) / 100) * 100
    sm[idx_s] = hz_to_cent_based_c4(sm[idx_s])
    sm[idx_s] = np.maximum(0, sm[idx_s] + shift)
    sm[idx_s] = cent_to_hz_based_c4(sm[idx_s])
    return sm
