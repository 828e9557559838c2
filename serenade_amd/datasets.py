"""FeatsDataset — decode-time feature reader with the semantics of
serenade/datasets/audio_mel_dataset.py:20-141 (keys, normalisation, returned dict)."""
import logging
import os

import numpy as np

from .utils.io import find_files, read_feats


class FeatsDataset(object):
    def __init__(self, root_dir, audio_query="*.h5", scaler=None, return_utt_id=False, allow_cache=False,
                 score_type="est_lf0_score", logmel_type="logmel"):
        files = sorted(find_files(root_dir, audio_query))
        if len(files) == 0 and audio_query == "*.h5":
            files = sorted(find_files(root_dir, "*.npz"))
        assert len(files) != 0, f"Not found any audio files in ${root_dir}."
        logging.info(f"score type: {score_type}")
        self.audio_files = files
        self.utt_ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
        self.scaler, self.return_utt_id = scaler, return_utt_id
        self.score_type, self.logmel_type = score_type, logmel_type
        self.allow_cache = allow_cache
        self.caches = [() for _ in files] if allow_cache else None

    def __len__(self):
        return len(self.audio_files)

    def __getitem__(self, idx):
        if self.allow_cache and len(self.caches[idx]) != 0:
            return self.caches[idx]
        f = self.audio_files[idx]
        audio, hubert = read_feats(f, "wave"), read_feats(f, "hubert")
        logmel, score = read_feats(f, self.logmel_type), read_feats(f, self.score_type)
        midi, loud, lf0 = read_feats(f, "midi"), read_feats(f, "loud"), read_feats(f, "f0")
        s = self.scaler
        if s is not None:
            logmel = (logmel - s["logmel"].mean_) / s["logmel"].scale_
            hubert = (hubert - s["hubert"].mean_) / s["hubert"].scale_
            score = (score - s["score"].data_min_) / (s["score"].data_max_ - s["score"].data_min_)
            loud = (loud - s["loud"].data_min_) / (s["loud"].data_max_ - s["loud"].data_min_)
            if np.isnan(logmel).any():
                logging.info(f"contains nan: {self.utt_ids[idx]}")
        items = {"audio": audio, "logmel": logmel, "hubert": hubert, "loud": loud, "score": score, "midi": midi,
                 "lf0": lf0}
        if self.return_utt_id:
            items["utt_id"] = self.utt_ids[idx]
        if self.allow_cache:
            self.caches[idx] = items
        return items

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
