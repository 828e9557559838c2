import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, serenade_amd
from serenade_amd import models
from serenade_amd.utils.synth import synth_inputs
B, T = 8, 4096
dev = torch.device("cuda:0")
d = synth_inputs(B, T, T_ref=256, seed=1235)
g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}
serenade_amd.set_precision("fp32")
for mb in (160, 320, 640, 1280, 2560):
    models.S_BUDGET = mb << 20
    model, voc, _, _ = bench.build_models(dev)
    def step():
        mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"], g["ref_logmel"], g["ref_midi"], g["ref_lft"], n_timesteps=10, noise=g["z"])
        return voc.decode_batch(mel)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): step()
    torch.cuda.synchronize()
    print(mb, "MiB:", round((time.perf_counter() - t0) / 2 * 1e3, 1), "ms", flush=True)
    del model, voc
    torch.cuda.empty_cache()
