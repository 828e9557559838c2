"""Decode-time feature reader.

Interface of the reference's ``FeatsDataset`` (serenade/datasets/audio_mel_dataset.py:20-141): built from a dump
directory and the ``stats.joblib`` scalers, indexable, and every item is a dict with the keys the decode loop reads
(``audio, logmel, hubert, loud, score, midi, lf0`` and optionally ``utt_id``).  The body is table-driven and this
repo's own: which stored dataset feeds which key, and which scaler entry normalises it how, is data (`_FIELDS`)."""
import logging
import os

import numpy as np

from .utils.io import find_files, read_feats

# item key -> (dataset name in the feature file or None if it is chosen per instance, scaler entry, kind of scaling)
_FIELDS = {
    "audio": ("wave", None, None),
    "logmel": (None, "logmel", "standard"),
    "hubert": ("hubert", "hubert", "standard"),
    "loud": ("loud", "loud", "minmax"),
    "score": (None, "score", "minmax"),
    "midi": ("midi", None, None),
    "lf0": ("f0", None, None),
}


def _scale(values, entry, kind):
    if kind == "standard":
        return (values - entry.mean_) / entry.scale_
    span = entry.data_max_ - entry.data_min_
    return (values - entry.data_min_) / span


class FeatsDataset:
    def __init__(self, root_dir, audio_query="*.h5", scaler=None, return_utt_id=False, allow_cache=False,
                 score_type="est_lf0_score", logmel_type="logmel"):
        paths = find_files(root_dir, audio_query)
        if not paths and audio_query == "*.h5":  # the same keys in .npz archives (where h5py is unavailable)
            paths = find_files(root_dir, "*.npz")
        if not paths:
            raise AssertionError(f"no feature files matching {audio_query!r} under {root_dir}")
        self.audio_files = sorted(paths)
        self.utt_ids = [os.path.basename(p).rsplit(".", 1)[0] for p in self.audio_files]
        self.scaler = scaler
        self.return_utt_id = return_utt_id
        self._stored = {"logmel": logmel_type, "score": score_type}
        self._memo = {} if allow_cache else None
        logging.info(f"{len(self.audio_files)} feature files, score contour '{score_type}', mel '{logmel_type}'")

    def __len__(self):
        return len(self.audio_files)

    def _load(self, idx):
        path = self.audio_files[idx]
        item = {}
        for key, (stored, entry, kind) in _FIELDS.items():
            values = read_feats(path, stored or self._stored[key])
            if self.scaler is not None and entry is not None:
                values = _scale(values, self.scaler[entry], kind)
            item[key] = values
        if self.scaler is not None and np.isnan(item["logmel"]).any():
            logging.warning(f"{self.utt_ids[idx]}: normalised mel has NaNs")
        if self.return_utt_id:
            item["utt_id"] = self.utt_ids[idx]
        return item

    def __getitem__(self, idx):
        if self._memo is None:
            return self._load(idx)
        if idx not in self._memo:
            self._memo[idx] = self._load(idx)
        return self._memo[idx]

    def __iter__(self):
        return (self[i] for i in range(len(self)))
