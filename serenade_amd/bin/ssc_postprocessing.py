#!/usr/bin/env python3
"""`serenade-postprocessing` on the MI355X: stage 9 of the recipe (egs/gtsinger/ssc1/run.sh:302-315), the interface of
serenade/bin/ssc_postprocessing.py with the WORLD re-analysis AND the SiFiGAN generator on the GPU.

    serenade-postprocessing generator=sifigan in_dir=DIR stats=stats.joblib checkpoint_path=model.pkl [key=value ...]

Options are hydra-style `key=value` overrides of the defaults below (the values of
serenade/bin/sifigan_config/ssc_postprocessing.yaml and generator/sifigan.yaml; nested keys as `generator.channels=256`;
hydra itself is not needed).  For every `*.wav` under in_dir whose name contains neither `_reference` nor `_gt`, the
transposed F0 contour is read from the `lf0` dataset the decode CLI left beside it (`.h5`, or `.npz` where h5py is
unavailable), the waveform is re-analysed (serenade_amd.world.Analyzer: CheapTrick -> mel-cepstrum, D4C -> band
aperiodicity, continuous F0, dilated factors, sine excitation) and `NAME_sifigan.wav` (PCM_16) is written next to it.
Files without an F0 contour or without a voiced frame are skipped, as in the reference.

Not reproduced: `pw.harvest` (:147-153) -- its F0 track is discarded by the reference, only its frame count and time
axis are used, and those are closed-form; `librosa.resample` for inputs at another rate (the decode CLI writes at
`sample_rate`): such files raise.  Under torchrun the file list is split over the ranks (no collective)."""
import copy
import glob
import logging
import os
import sys
import time

import numpy as np
import torch
import yaml

from serenade_amd import parallel, sifigan, world
from serenade_amd.utils.io import read_feats, read_wav, write_wav_pcm16

logger = logging.getLogger(__name__)

DEFAULTS = {
    "in_dir": None, "out_dir": None, "stats": None, "checkpoint_path": None, "f0_factors": [1.00], "seed": 100,
    "sample_rate": 24000, "frame_period": 5, "f0_floor": 100, "f0_ceil": 840, "mcep_dim": 39, "mcap_dim": 19,
    "aux_feats": ["mcep", "bap"], "dense_factors": [0.5, 1, 4, 8], "df_f0_type": "cf0", "sine_amp": 0.1,
    "noise_amp": 0.003, "sine_f0_type": "cf0", "signal_types": ["sine"],
    "generator": {k: (list(v) if isinstance(v, tuple) else copy.deepcopy(v)) for k, v in sifigan.DEFAULT_PARAMS.items()},
}
GENERATORS = ("sifigan",)


def parse_overrides(argv):
    cfg = copy.deepcopy(DEFAULTS)
    for arg in argv:
        if "=" not in arg:
            raise SystemExit(f"expected key=value, got {arg!r}")
        key, text = arg.split("=", 1)
        if key == "generator":  # hydra config-group choice
            if text not in GENERATORS:
                raise SystemExit(f"unknown generator {text!r} (available: {', '.join(GENERATORS)})")
            continue
        node = cfg
        *parents, leaf = key.split(".")
        for part in parents:
            if not isinstance(node.get(part), dict):
                raise SystemExit(f"unknown option {key!r}")
            node = node[part]
        if leaf not in node:
            raise SystemExit(f"unknown option {key!r}")
        node[leaf] = yaml.safe_load(text) if text != "" else None
    return cfg


class PostJob:
    def __init__(self, cfg):
        self.cfg = cfg
        for k in ("in_dir", "stats", "checkpoint_path"):
            if not cfg[k]:
                raise SystemExit(f"{k}= is required")
        if not torch.cuda.is_available():
            raise RuntimeError("serenade-postprocessing (MI355X build) needs a GPU: there is no CPU path")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        self.device = torch.device("cuda", torch.cuda.current_device())
        np.random.seed(cfg["seed"])
        torch.manual_seed(cfg["seed"])
        gen_cfg = {k: v for k, v in cfg["generator"].items() if k != "_target_"}
        self.model = sifigan.SiFiGANGenerator(**gen_cfg)
        state = torch.load(cfg["checkpoint_path"], map_location="cpu")
        self.model.load_state_dict(state["model"]["generator"])
        logger.info(f"Loaded model parameters from {cfg['checkpoint_path']}.")
        self.model.remove_weight_norm()
        self.model = self.model.eval().to(self.device)
        from joblib import load
        self.analyzer = world.Analyzer(
            sample_rate=cfg["sample_rate"], frame_period=cfg["frame_period"], mcep_dim=cfg["mcep_dim"],
            dense_factors=cfg["dense_factors"], upsample_scales=gen_cfg["upsample_scales"],
            df_f0_type=cfg["df_f0_type"], sine_amp=cfg["sine_amp"], noise_amp=cfg["noise_amp"],
            sine_f0_type=cfg["sine_f0_type"], signal_types=cfg["signal_types"], aux_feats=cfg["aux_feats"],
            scaler=load(cfg["stats"]), pcm16=False)  # the samples read from a PCM_16 file are already on its grid

    def files(self):
        paths = glob.glob(os.path.join(self.cfg["in_dir"], "**", "*.wav"), recursive=True)
        return sorted(p for p in paths if "_reference" not in p and "_gt" not in p)

    def f0_of(self, wav_file):
        stem = wav_file[:-len(".wav")]
        for ext in (".h5", ".npz"):
            if os.path.exists(stem + ext):
                return read_feats(stem + ext, "lf0")
        return None

    def process(self, wav_file):
        x, sr = read_wav(wav_file)
        if np.ndim(x) != 1:
            raise ValueError(f"{wav_file}: {np.shape(x)[1]} channels; the analysis expects mono audio")
        if sr != self.cfg["sample_rate"]:
            raise NotImplementedError(f"{wav_file}: {sr} Hz, expected {self.cfg['sample_rate']} (no resampler built)")
        f0 = self.f0_of(wav_file)
        if f0 is None:
            print(f"No h5 file containing f0 found for {wav_file}")
            return 0
        wave = torch.from_numpy(np.asarray(x, dtype=np.float32)).to(self.device).view(1, -1)
        in_signal, c, dfs, feats = self.analyzer(wave, [wave.size(1)], [np.asarray(f0)])
        if int(feats["ok"][0]) == 0:
            logger.warning(f"{wav_file}: all of the f0 values are 0.")
            return 0
        y = self.model(in_signal, c, dfs)[0]
        write_wav_pcm16(wav_file[:-len(".wav")] + "_sifigan.wav", y.view(-1).cpu().numpy(), self.cfg["sample_rate"])
        return c.size(-1)

    def run(self):
        rank, n_ranks = parallel.rank_world()
        files = self.files()
        lo, hi = parallel.shard_range(len(files), rank, n_ranks)
        logger.info(f"Processing {self.cfg['in_dir']}: {hi - lo} of {len(files)} files on rank {rank}/{n_ranks}")
        frames, t0 = 0, time.time()
        for key in ("mcap_dim", "f0_floor", "f0_ceil"):  # accepted like the reference's yaml, not used by this data flow
            if self.cfg[key] != DEFAULTS[key]:
                logger.warning(f"{key}={self.cfg[key]} has no effect here: the F0 comes from the decode CLI's lf0 and `mcap` "
                               "is never consumed with aux_feats [mcep, bap] (ssc_postprocessing.py:147-170)")
        with torch.no_grad():
            for wav_file in files[lo:hi]:
                logger.info(f"Start processing {wav_file}")
                try:
                    frames += self.process(wav_file)
                except (ValueError, NotImplementedError, OSError) as e:  # one bad file must not end the directory run
                    logger.error(f"skipped {wav_file}: {e}")
        torch.cuda.synchronize()
        dt = max(time.time() - t0, 1e-9)
        logger.info(f"rank {rank}/{n_ranks}: {frames} analysis frames in {dt:.2f} s = {frames / dt:.1f} frames/s")
        return frames


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(levelname)s][%(module)s | %(lineno)s] %(message)s")
    cfg = parse_overrides(sys.argv[1:] if argv is None else argv)
    return PostJob(cfg).run()


if __name__ == "__main__":
    main()
