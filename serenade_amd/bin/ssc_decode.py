#!/usr/bin/env python3
"""`serenade-decode` for MI355X.

Drop-in for the reference's decode entry point (interface: serenade/bin/ssc_decode.py:192-253 flags,
:361-366/:395-400/:444-455 output files).  The command line, the files read and the files written are the
reference's; the program itself is organised as a small job object around the HIP-backed model and vocoder:

    python -m serenade_amd.bin.ssc_decode --dumpdir DIR --stats stats.joblib --ref-dict refs.json \
        --outdir OUT --checkpoint checkpoint.pkl [--config config.yml] [--verbose 1]

Per source utterance `U` and reference style `S` it writes `U_gt.wav`, `00_S_reference.wav`, `U_S.wav` (16-bit
PCM) and the transposed F0 contour as dataset `lf0` of `U_S.h5` (`.npz` when the dump is `.npz`).  By default every
(utterance, style) pair is one B = 1 call, like the reference: GroupNorm statistics of the estimator run over an
item's padded length (decoder.py:71-77), so naive padding into a batch would change the output.  `--batch-styles`
converts all styles of an utterance in one *exact* ragged batch (`Serenade.inference_ragged`: per-item GroupNorm
statistics, per-item reflection padding, per-item prompt / source offsets), whose results equal the loop's;
`--batch-utterances N` extends the batch over N source utterances of different lengths.  With
`torchrun --nproc-per-node N` the utterance list is split contiguously over the ranks (one GPU each).
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
from serenade_amd import parallel
from serenade_amd.datasets import FeatsDataset
from serenade_amd.utils.f0 import linear_midi_shift
from serenade_amd.utils.io import read_feats, write_feats, write_wav_pcm16
from serenade_amd.vocoder import Vocoder

STYLES = ("Breathy", "Falsetto", "Pharyngeal", "Mixed_Voice")


def _sibling_dump(dirname):
    """the other half of a two-way dump split (`dump.1` <-> `dump.2`), or None"""
    for a, b in (("dump.2", "dump.1"), ("dump.1", "dump.2")):
        if a in dirname:
            return dirname.replace(a, b)
    return None


def get_random_ref_style(dumpdir, utt_id, ext="h5"):
    """{style: feature file} with one randomly drawn prompt per singing style, taken from the same language and
    singer as `utt_id` (file names are `<lang>_<singer>_<song>_<style>_Group_<n>`); styles without any candidate in
    `dumpdir` or its sibling split are left out."""
    lang, singer = utt_id.split("_")[:2]
    chosen = {}
    for style in STYLES:
        pattern = f"{lang}_{singer}_*_{style}_Group_*.{ext}"
        found = glob.glob(os.path.join(dumpdir, pattern))
        if not found and _sibling_dump(dumpdir) is not None:
            found = glob.glob(os.path.join(_sibling_dump(dumpdir), pattern))
        if found:
            chosen[style] = np.random.choice(found)
    logging.info(f"prompt per style: {chosen}")
    return chosen


def build_parser():
    p = argparse.ArgumentParser(description="Singing-style conversion of a feature dump with a trained Serenade "
                                            "model and its vocoder (MI355X build).")
    p.add_argument("--config", default=None, type=str,
                   help="training configuration (YAML); defaults to config.yml beside --checkpoint")
    p.add_argument("--feats-scp", "--scp", default=None, type=str,
                   help="Kaldi-style feats.scp listing the inputs (give exactly one of --feats-scp / --dumpdir)")
    p.add_argument("--dumpdir", default=None, type=str,
                   help="directory of per-utterance feature files (give exactly one of --feats-scp / --dumpdir)")
    p.add_argument("--stats", type=str, required=True,
                   help="joblib file of the fitted feature scalers; its log-mel scaler de-normalises the output")
    p.add_argument("--ref-dict", type=str, default=None,
                   help="JSON map style -> prompt feature file; without it one prompt per style is drawn at random")
    p.add_argument("--outdir", type=str, required=True, help="where the converted audio and F0 files go")
    p.add_argument("--checkpoint", type=str, required=True, help="model checkpoint (its ['model'] entry is loaded)")
    p.add_argument("--verbose", type=int, default=1, help="0: warnings only, 1: info, 2+: debug (default 1)")
    p.add_argument("--batch-styles", action="store_true",
                   help="(MI355X build only) convert all reference styles of an utterance in ONE ragged batch "
                        "(Serenade.inference_ragged: every item is computed exactly as its own B = 1 call, so the "
                        "outputs equal the default style-by-style loop) -- fills the GPU instead of running B = 1")
    p.add_argument("--batch-utterances", type=int, default=1,
                   help="(MI355X build only) convert this many source utterances, with all their styles, per exact "
                        "ragged batch (implies --batch-styles); outputs equal the one-by-one loop")
    p.add_argument("--sifigan-checkpoint", type=str, default=None,
                   help="(MI355X build only) also run the recipe's stage 9 in this process: re-analyse every converted "
                        "waveform on the GPU (serenade_amd.world.Analyzer) and write NAME_sifigan.wav from this SiFiGAN "
                        "generator checkpoint -- the same samples `serenade-postprocessing` would write from the wav "
                        "and lf0 files, without the round trip through them")
    p.add_argument("--sifigan-stats", type=str, default=None,
                   help="joblib file with the `mcep` / `bap` scalers of the SiFiGAN model (with --sifigan-checkpoint)")
    p.add_argument("--sifigan-noise-amp", type=float, default=0.003,
                   help="noise_amp of ssc_postprocessing.yaml (with --sifigan-checkpoint)")
    return p


def _setup_logging(verbose):
    level = logging.WARN if verbose <= 0 else logging.INFO if verbose == 1 else logging.DEBUG
    logging.basicConfig(level=level, format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    if verbose <= 0:
        logging.warning("INFO and DEBUG messages are suppressed")


class DecodeJob:
    """Everything one decode run needs: merged config, scalers, vocoder, model, dataset."""

    def __init__(self, args):
        self.args = args
        if torch.cuda.is_available():  # one process per GPU under torchrun; plain runs use device 0
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.device = torch.device("cpu")
        cfg_path = args.config or os.path.join(os.path.dirname(args.checkpoint), "config.yml")
        with open(cfg_path) as f:
            self.config = yaml.load(f, Loader=yaml.Loader)
        self.config.update(vars(args))  # command line wins, as in the reference (ssc_decode.py:287-289)
        self.styles, self._sharded = None, False
        if args.ref_dict is not None:
            with open(args.ref_dict) as f:
                self.styles = json.load(f)
        else:
            logging.info("no --ref-dict: drawing one random prompt per style")
        for k, v in self.config.items():
            logging.info(f"{k} = {v}")

        from joblib import load
        self.scaler = load(args.stats)
        mel_scaler = self.scaler["logmel"]
        self.config["trg_stats"] = {"mean": mel_scaler.mean_, "scale": mel_scaler.scale_}
        voc = self.config["vocoder"]
        self.vocoder = Vocoder(voc["checkpoint"], voc["config"], voc["stats"], self.device,
                               trg_stats=self.config["trg_stats"])
        if (args.feats_scp is None) == (args.dumpdir is None):
            raise ValueError("Please specify either --dumpdir or --feats-scp.")
        self.dataset = FeatsDataset(root_dir=args.dumpdir, scaler=self.scaler, score_type="est_lf0_score",
                                    return_utt_id=True, allow_cache=self.config.get("allow_cache", False))
        logging.info(f"{len(self.dataset)} utterances to convert")
        cls = getattr(serenade_amd.models, self.config["model_type"])  # the reference's plugin lookup (:337)
        self.model = cls(**self.config["model_params"])
        self.model.load_state_dict(torch.load(args.checkpoint, map_location="cpu")["model"])
        self.model = self.model.eval().to(self.device)
        logging.info(f"model weights: {args.checkpoint}")
        self.sr = self.config["sampling_rate"]
        self.ext = os.path.splitext(self.dataset.audio_files[0])[1].lstrip(".")
        self.post = None
        if args.sifigan_checkpoint is not None:
            if args.sifigan_stats is None:
                raise ValueError("--sifigan-checkpoint needs --sifigan-stats")
            from serenade_amd.bin.ssc_postprocessing import DEFAULTS
            from serenade_amd import sifigan, world
            gen = sifigan.SiFiGANGenerator(**DEFAULTS["generator"])
            gen.load_state_dict(torch.load(args.sifigan_checkpoint, map_location="cpu")["model"]["generator"])
            gen.remove_weight_norm()
            self.post = (world.Analyzer(sample_rate=self.sr, scaler=load(args.sifigan_stats), pcm16=True,
                                        noise_amp=args.sifigan_noise_amp),
                         gen.eval().to(self.device))

    # ---- tensors -------------------------------------------------------------------------------------------
    def _t(self, a):
        return torch.tensor(np.asarray(a), dtype=torch.float).to(self.device)

    def _standard(self, a, key):
        s = self.scaler[key]
        return (a - s.mean_) / s.scale_

    def _minmax(self, a, key):
        s = self.scaler[key]
        return (a - s.data_min_) / (s.data_max_ - s.data_min_)

    def prompt(self, path):
        """normalised prompt tensors + raw prompt audio / F0 of one reference feature file"""
        raw = {k: read_feats(path, k) for k in ("hubert", "logmel", "loud", "wave", "est_lf0_score", "f0")}
        cvec = self._t(self._standard(raw["hubert"], "hubert")).unsqueeze(0)
        return dict(cvec=cvec, lens=torch.tensor([cvec.size(1)], dtype=torch.long),
                    mel=self._t(self._standard(raw["logmel"], "logmel")).unsqueeze(0),
                    score=self._t(self._minmax(raw["est_lf0_score"], "score")).view(1, -1, 1),
                    loud=self._t(self._minmax(raw["loud"], "loud")).unsqueeze(0), wave=raw["wave"], f0=raw["f0"])

    # ---- the loop ------------------------------------------------------------------------------------------
    def _jobs(self, item):
        """host side of one source utterance: writes the ground-truth / reference audio, returns the (utterance,
        style) conversions to run: (utt, style, x, score, loud, prompt tensors, transposed F0)"""
        out = self.args.outdir
        utt = item["utt_id"]
        logging.info(f"utterance {utt}")
        write_wav_pcm16(os.path.join(out, f"{utt}_gt.wav"), item["audio"], self.sr)
        x = self._t(item["hubert"]).unsqueeze(0)
        score, loud = self._t(item["score"]).unsqueeze(0), self._t(item["loud"]).unsqueeze(0)
        if self.styles is None:  # drawn once, for the first utterance, then kept (ssc_decode.py:375-376)
            self.styles = get_random_ref_style(self.args.dumpdir, utt, self.ext)
        jobs = []
        for style, path in self.styles.items():
            if style in utt:  # a prompt of the utterance's own style would be a reconstruction
                continue
            logging.info(f"  style {style}")
            ref = self.prompt(path)
            if not self._sharded:  # several ranks: rank 0 wrote every prompt's audio once, up front (run())
                write_wav_pcm16(os.path.join(out, f"00_{style}_reference.wav"), ref["wave"], self.sr)
            # NB: linear_midi_shift edits item["lf0"] in place, so later styles start from the shifted contour --
            # the reference behaves the same way (ssc_decode.py:424); snapshot what it would write for this style
            lf0 = linear_midi_shift(item["lf0"], ref["f0"]).astype(np.float32)
            jobs.append((utt, style, x, score, loud, ref, lf0))
        return jobs

    def _run_jobs(self, jobs, batched):
        """convert and write a list of jobs; batched: ONE exact ragged batch for the model (every item = its own B = 1
        call) and one vocoder batch per distinct length (HiFi-GAN has no cross-item arithmetic, but its edge padding
        sits at the tensor's end, so only equal lengths share a batch)"""
        if not jobs:
            return 0
        out = self.args.outdir
        if batched and len(jobs) > 1:
            mels = self.model.inference_ragged(
                [(x[0], sc[0], ld[0], r["cvec"][0], r["mel"][0], r["score"][0], r["loud"][0])
                 for _, _, x, sc, ld, r, _ in jobs])
            waves = [None] * len(jobs)
            by_len = {}
            for i, m in enumerate(mels):
                by_len.setdefault(m.shape[0], []).append(i)
            for idx in by_len.values():
                for i, w in zip(idx, self.vocoder.decode_batch(torch.stack([mels[i] for i in idx]))):
                    waves[i] = w
        else:
            waves = []
            for _, _, x, sc, ld, r, _ in jobs:
                lengths = torch.tensor([x.shape[1]], dtype=torch.long)
                mel = self.model.inference(x, lengths, sc, ld, r["cvec"], r["lens"], r["mel"], r["score"], r["loud"])
                waves.append(self.vocoder.decode(mel.squeeze(0) if mel.dim() == 3 else mel)[0])
        done = 0
        for (utt, style, x, _, _, _, lf0), wave in zip(jobs, waves):
            write_feats(os.path.join(out, f"{utt}_{style}.{self.ext}"), "lf0", lf0)
            write_wav_pcm16(os.path.join(out, f"{utt}_{style}.wav"), wave.cpu().numpy(), self.sr)
            if self.post is not None:  # stage 9 without leaving the GPU: waveform -> WORLD features -> SiFiGAN
                analyzer, gen = self.post
                w = wave.reshape(1, -1).to(torch.float32)
                in_signal, c, dfs, feats = analyzer(w, [w.size(1)], [np.asarray(lf0)])
                if int(feats["ok"][0]):
                    y = gen(in_signal, c, dfs)[0]
                    write_wav_pcm16(os.path.join(out, f"{utt}_{style}_sifigan.wav"), y.view(-1).cpu().numpy(), self.sr)
                else:
                    logging.warning(f"{utt}_{style}: all of the f0 values are 0, no SiFiGAN output")
            done += x.shape[1]
        return done

    def convert(self, item):
        """all styles of one source utterance; returns the number of converted frames"""
        return self._run_jobs(self._jobs(item), self.args.batch_styles)

    def run(self):
        rank, world = parallel.rank_world()
        self._sharded = world > 1
        if self._sharded:
            # one run = one set of prompts: a per-rank random draw would convert with different prompts on different
            # ranks and leave whichever 00_*_reference.wav was written last (the reference draws once per run)
            if self.styles is None:
                raise ValueError("several ranks need --ref-dict: random prompts are drawn per process")
            if rank == 0:
                for style, path in self.styles.items():
                    write_wav_pcm16(os.path.join(self.args.outdir, f"00_{style}_reference.wav"),
                                    read_feats(path, "wave"), self.sr)
        lo, hi = parallel.shard_range(len(self.dataset), rank, world)
        frames, t0 = 0, time.time()
        n_utt = max(1, int(self.args.batch_utterances))
        with torch.no_grad():
            if n_utt == 1:
                for i in range(lo, hi):
                    frames += self.convert(self.dataset[i])
            else:  # several utterances (all their styles) per exact ragged batch
                for i0 in range(lo, hi, n_utt):
                    jobs = []
                    for i in range(i0, min(i0 + n_utt, hi)):
                        jobs += self._jobs(self.dataset[i])
                    frames += self._run_jobs(jobs, True)
        dt = max(time.time() - t0, 1e-9)
        logging.info(f"rank {rank}/{world}: {frames} frames in {dt:.2f} s = {frames / dt:.1f} frames/s")
        return frames


def main(argv=None):
    args = build_parser().parse_args(argv)
    _setup_logging(args.verbose)
    os.makedirs(args.outdir, exist_ok=True)
    DecodeJob(args).run()


if __name__ == "__main__":
    main()
