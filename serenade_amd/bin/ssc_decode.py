#!/usr/bin/env python3
"""`serenade-decode` on MI355X — same command line, inputs and outputs as serenade/bin/ssc_decode.py:190-455:

    python -m serenade_amd.bin.ssc_decode --dumpdir DIR --stats stats.joblib --ref-dict refs.json \
        --outdir OUT --checkpoint checkpoint.pkl [--config config.yml] [--verbose 1]

writes `{utt}_gt.wav`, `00_{style}_reference.wav`, `{utt}_{style}.wav` (PCM_16) and `{utt}_{style}.h5["lf0"]`.
The model and vocoder are the HIP-backed classes of serenade_amd; feature files may be .h5 (h5py) or .npz.
"""
import argparse
import glob
import json
import logging
import os
import time

import numpy as np
import torch
import yaml

import serenade_amd.models
from serenade_amd.datasets import FeatsDataset
from serenade_amd.utils.f0 import linear_midi_shift
from serenade_amd.utils.io import read_feats, write_feats, write_wav_pcm16
from serenade_amd.vocoder import Vocoder


def get_random_ref_style(dumpdir, utt_id, ext="h5"):
    """One random reference file per singing style for the speaker of `utt_id` (ssc_decode.py:157-187)."""
    dirname = dumpdir
    ln, spk = utt_id.split("_")[:2]
    ref_dict = {}
    for style in ["Breathy", "Falsetto", "Pharyngeal", "Mixed_Voice"]:
        name = f"{ln}_{spk}_*_{style}_Group_*.{ext}"
        files = glob.glob(os.path.join(dirname, name))
        if not files:
            other = None
            if "dump.2" in dirname:
                other = dirname.replace("dump.2", "dump.1")
            elif "dump.1" in dirname:
                other = dirname.replace("dump.1", "dump.2")
            if other is not None:
                files = glob.glob(os.path.join(other, name))
        if files:
            ref_dict[style] = np.random.choice(files)
    logging.info(f"Using reference styles: {ref_dict}")
    return ref_dict


def build_parser():
    p = argparse.ArgumentParser(description="Decode with trained SSC model (See detail in bin/ssc_decode.py).")
    p.add_argument("--config", default=None, type=str,
                   help="yaml format configuration file. if not explicitly provided, it will be searched in the "
                        "checkpoint directory. (default=None)")
    p.add_argument("--feats-scp", "--scp", default=None, type=str,
                   help="kaldi-style feats.scp file. you need to specify either feats-scp or dumpdir.")
    p.add_argument("--dumpdir", default=None, type=str,
                   help="directory including feature files. you need to specify either feats-scp or dumpdir.")
    p.add_argument("--stats", type=str, required=True, help="stats file for target denormalization.")
    p.add_argument("--ref-dict", type=str, default=None, help="yaml format file containing reference styles.")
    p.add_argument("--outdir", type=str, required=True, help="directory to save generated speech.")
    p.add_argument("--checkpoint", type=str, required=True, help="checkpoint file to be loaded.")
    p.add_argument("--verbose", type=int, default=1, help="logging level. higher is more logging. (default=1)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    fmt = "%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s"
    level = logging.DEBUG if args.verbose > 1 else logging.INFO if args.verbose > 0 else logging.WARN
    logging.basicConfig(level=level, format=fmt)
    if args.verbose <= 0:
        logging.warning("Skip DEBUG/INFO messages")
    os.makedirs(args.outdir, exist_ok=True)

    device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    if args.config is None:
        args.config = os.path.join(os.path.dirname(args.checkpoint), "config.yml")
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.Loader)
    config.update(vars(args))
    ref_dict = None
    if args.ref_dict is not None:
        with open(args.ref_dict, "r") as f:
            ref_dict = json.load(f)
    else:
        logging.info("No reference dictionary provided, using random reference styles.")
    for key, value in config.items():
        logging.info(f"{key} = {value}")

    from joblib import load
    scaler = load(args.stats)
    config["trg_stats"] = {"mean": scaler["logmel"].mean_, "scale": scaler["logmel"].scale_}
    vocoder = Vocoder(config["vocoder"]["checkpoint"], config["vocoder"]["config"], config["vocoder"]["stats"], device,
                      trg_stats=config["trg_stats"])
    if (args.feats_scp is not None and args.dumpdir is not None) or (args.feats_scp is None and args.dumpdir is None):
        raise ValueError("Please specify either --dumpdir or --feats-scp.")
    dataset = FeatsDataset(root_dir=args.dumpdir, scaler=scaler, score_type="est_lf0_score", return_utt_id=True,
                           allow_cache=config.get("allow_cache", False))
    logging.info(f"The number of features to be decoded = {len(dataset)}.")

    model_class = getattr(serenade_amd.models, config["model_type"])
    model = model_class(**config["model_params"])
    model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"])
    model = model.eval().to(device)
    logging.info(f"Loaded model parameters from {args.checkpoint}.")
    sr = config["sampling_rate"]
    fext = os.path.splitext(dataset.audio_files[0])[1].lstrip(".")

    def dev(a, shape=None):
        t = torch.tensor(np.asarray(a), dtype=torch.float).to(device)
        return t.view(*shape) if shape is not None else t

    n_frames, t_start = 0, time.time()
    with torch.no_grad():
        for batch in dataset:
            utt_id = batch["utt_id"]
            logging.info(f"Decoding {utt_id}")
            lf0 = batch["lf0"]
            write_wav_pcm16(os.path.join(args.outdir, f"{utt_id}_gt.wav"), batch["audio"], sr)
            x = dev(batch["hubert"]).unsqueeze(0)
            lengths = torch.tensor([x.shape[1]], dtype=torch.long)
            scores = dev(batch["score"]).unsqueeze(0)
            lfts = dev(batch["loud"]).unsqueeze(0)
            if ref_dict is None:
                ref_dict = get_random_ref_style(args.dumpdir, utt_id, fext)
            for style, ref_path in ref_dict.items():
                if style in utt_id:  # avoid reconstruction
                    continue
                logging.info(f"Processing reference style: {style}")
                ref_cvec, ref_mel = read_feats(ref_path, "hubert"), read_feats(ref_path, "logmel")
                ref_lft, ref_wave = read_feats(ref_path, "loud"), read_feats(ref_path, "wave")
                ref_score, ref_lf0 = read_feats(ref_path, "est_lf0_score"), read_feats(ref_path, "f0")
                write_wav_pcm16(os.path.join(args.outdir, f"00_{style}_reference.wav"), ref_wave, sr)
                ref_cvec = dev((ref_cvec - scaler["hubert"].mean_) / scaler["hubert"].scale_).unsqueeze(0)
                ref_mel = dev((ref_mel - scaler["logmel"].mean_) / scaler["logmel"].scale_).unsqueeze(0)
                ref_lns = torch.tensor([ref_cvec.size(1)], dtype=torch.long)
                ref_score = dev((ref_score - scaler["score"].data_min_) /
                                (scaler["score"].data_max_ - scaler["score"].data_min_), (1, -1, 1))
                ref_lft = dev((ref_lft - scaler["loud"].data_min_) /
                              (scaler["loud"].data_max_ - scaler["loud"].data_min_)).unsqueeze(0)
                shifted_lf0 = linear_midi_shift(lf0, ref_lf0)
                mel_ = model.inference(x, lengths, scores, lfts, ref_cvec, ref_lns, ref_mel, ref_score, ref_lft)
                wave, _ = vocoder.decode(mel_.squeeze(0))
                outname = f"{utt_id}_{style}"
                write_feats(os.path.join(args.outdir, f"{outname}.{fext}"), "lf0", shifted_lf0.astype(np.float32))
                write_wav_pcm16(os.path.join(args.outdir, f"{outname}.wav"), wave.cpu().numpy(), sr)
                n_frames += x.shape[1]
    dt = time.time() - t_start
    logging.info(f"Converted {n_frames} frames in {dt:.2f} s ({n_frames / max(dt, 1e-9):.1f} frames/s).")


if __name__ == "__main__":
    main()
