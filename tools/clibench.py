#!/usr/bin/env python3
"""Developer tool (GPU): the decode CLI end to end on a synthetic dump with the full-size model: the
reference's one-by-one loop against --batch-styles and --batch-utterances.  Wall time of `DecodeJob.run()` (feature
reads, HIP model + vocoder, PCM / F0 writes), second run of each mode (plans and weight planes warm)."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _shapes  # noqa: E402
from serenade_amd.bin import ssc_decode  # noqa: E402
from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict  # noqa: E402


def feats(rng, T):
    return dict(wave=(0.1 * rng.standard_normal(T * 240)).astype(np.float32),
                hubert=rng.standard_normal((T, 768)).astype(np.float32),
                logmel=rng.standard_normal((T, 80)).astype(np.float32),
                loud=rng.uniform(-40, 0, (T, 1)).astype(np.float32),
                est_lf0_score=rng.uniform(40, 70, (T, 1)).astype(np.float32),
                midi=rng.uniform(40, 70, (T, 1)).astype(np.float32),
                f0=np.where(rng.uniform(size=T) > 0.2, rng.uniform(150, 400, T), 0.0))


def main():
    from joblib import dump
    from sklearn.preprocessing import MinMaxScaler, StandardScaler
    lens = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,410,333,512,290,450,380,300").split(",")]
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        for sub in ("dump", "exp", "ref", "voc"):
            os.makedirs(os.path.join(d, sub))
        for i, T in enumerate(lens):
            np.savez(os.path.join(d, "dump", f"EN_spk1_song_Control_Group_{i:04d}.npz"), **feats(rng, T))
        styles = {}
        for s, R in zip(ssc_decode.STYLES, (256, 300, 220, 280)):
            styles[s] = os.path.join(d, "ref", s + ".npz")
            np.savez(styles[s], **feats(rng, R))
        dump({"logmel": StandardScaler().fit(rng.standard_normal((50, 80))),
              "hubert": StandardScaler().fit(rng.standard_normal((50, 768))),
              "score": MinMaxScaler().fit(np.array([[30.0], [80.0]])),
              "loud": MinMaxScaler().fit(np.array([[-50.0], [0.0]]))}, os.path.join(d, "stats.joblib"))
        torch.save({"model": fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**SERENADE_PARAMS)), seed=0)},
                   os.path.join(d, "exp", "checkpoint.pkl"))
        torch.save({"model": {"generator": fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**HIFIGAN_PARAMS, weight_norm=True)), seed=0)}},
                   os.path.join(d, "voc", "vocoder.pkl"))
        yaml.safe_dump({"generator_params": HIFIGAN_PARAMS, "sampling_rate": 24000, "format": "hdf5"},
                       open(os.path.join(d, "voc", "config.yml"), "w"))
        np.savez(os.path.join(d, "voc", "stats.npz"), mean=np.zeros(80, np.float32), scale=np.ones(80, np.float32))
        yaml.safe_dump({"model_type": "Serenade", "model_params": SERENADE_PARAMS, "sampling_rate": 24000,
                        "vocoder": {"checkpoint": os.path.join(d, "voc", "vocoder.pkl"),
                                    "config": os.path.join(d, "voc", "config.yml"),
                                    "stats": os.path.join(d, "voc", "stats.npz")}},
                       open(os.path.join(d, "exp", "config.yml"), "w"))
        json.dump(styles, open(os.path.join(d, "refs.json"), "w"))
        base = ["--dumpdir", os.path.join(d, "dump"), "--stats", os.path.join(d, "stats.joblib"), "--ref-dict",
                os.path.join(d, "refs.json"), "--checkpoint", os.path.join(d, "exp", "checkpoint.pkl"), "--verbose", "0"]
        res = {"utterance_frames": lens, "styles": 4, "conversions": 4 * len(lens)}
        for name, extra in (("loop", []), ("batch_styles", ["--batch-styles"]), ("batch_utterances_4", ["--batch-utterances", "4"])):
            args = ssc_decode.build_parser().parse_args(base + ["--outdir", os.path.join(d, name)] + extra)
            os.makedirs(args.outdir, exist_ok=True)
            job = ssc_decode.DecodeJob(args)
            job.run()  # warm: plans, weight planes
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frames = job.run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = {"seconds": dt, "frames_per_s": frames / dt}
        print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
