#!/usr/bin/env python3
"""Headline benchmark: converted source mel frames / second of the Serenade audio-infilling inference path
(GST style encoder -> content encoder -> 1-D UNet flow-matching Euler ODE -> HiFi-GAN vocoder) on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic batch per GPU: `Serenade.inference` (B=8 utterances of
T=1024 source frames with a 256-frame prompt, 10 Euler steps) + `Vocoder.decode_batch`, inputs already resident
in HBM; with N > 1 every rank converts its own batch (utterances shard embarrassingly, weak scaling) and the
converted waveforms are gathered on rank 0 over RCCL inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU, T_SRC, T_REF, N_EULER = 8, 1024, 256, 10
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense MFMA peak (one MFMA pass; split-bf16 needs three)


def algorithmic_flops(B, T, T_ref, n):
    """SURVEY.md section 8(d) per-utterance work: estimator 81.3e6*L + 24576*L^2 per call, encoder 11.32e6 per
    frame, HiFi-GAN 499e6 per source frame (all of it runs in the conv_gemm kernel; GST's 0.8 GFLOP does not)."""
    L = T + T_ref
    return B * (n * (81.3e6 * L + 24576.0 * L * L) + 11.32e6 * (T + T_ref) + 499e6 * T)


def build_models(dev):
    from serenade_amd import _shapes, models, vocoder
    from serenade_amd.utils.synth import HIFIGAN_PARAMS, SERENADE_PARAMS, fill_state_dict
    sd = fill_state_dict(_shapes.as_meta(_shapes.serenade_shapes(**SERENADE_PARAMS)), seed=0)
    model = models.Serenade(**SERENADE_PARAMS)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    gsd = fill_state_dict(_shapes.as_meta(_shapes.hifigan_shapes(**HIFIGAN_PARAMS, weight_norm=True)), seed=0)
    gen = vocoder.HiFiGANGenerator(**HIFIGAN_PARAMS)
    gen.load_state_dict(gsd)
    one = np.ones(80, dtype=np.float32)
    ident = {"mean": 0 * one, "scale": one}
    voc = vocoder.Vocoder.from_generator(gen, {"sampling_rate": 24000}, ident, dev, trg_stats=ident)
    return model, voc, sd, gsd


def cpu_baseline(sd, gsd):
    """The CPU oracle (a port of the reference's CPU path, pinned to its golden vectors) timed on this box's
    host cores on a bounded sample: ONE utterance of the same workload."""
    from oracle import serenade_oracle as O
    from serenade_amd.utils.synth import HIFIGAN_PARAMS, synth_inputs
    d = synth_inputs(1, T_SRC, T_ref=T_REF, seed=1235)
    threads = torch.get_num_threads()
    t1 = torch.ones(80)
    ident = {"mean": 0 * t1, "scale": t1}
    with torch.no_grad():
        t0 = time.time()
        mel = O.serenade_inference(sd, d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                                   d["ref_logmel"], d["ref_midi"], d["ref_lft"], d["z"], n_timesteps=N_EULER)
        O.vocoder_decode(gsd, mel, HIFIGAN_PARAMS, ident, ident)
        dt = time.time() - t0
    return {"value": T_SRC / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 utterance of the workload (T={T_SRC}, T_ref={T_REF}, {N_EULER} Euler steps + HiFi-GAN), "
                      f"fp32 torch CPU, {dt:.1f} s"}


def main():
    global B_PER_GPU, T_SRC, T_REF, N_EULER
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-fp32", action="store_true", help="profiling runs: time the headline split-bf16 mode only")
    ap.add_argument("--graphs", type=int, default=None, help="1/0: replay the plans as hipGraphs (default: library default)")
    # the north-star's other sizes (T in {256, 1024, 4096}, 20 Euler steps); the defaults are the headline workload
    ap.add_argument("--frames", type=int, default=T_SRC, help="source mel frames per utterance")
    ap.add_argument("--ref-frames", type=int, default=T_REF, help="prompt frames per utterance")
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="utterances per GPU")
    ap.add_argument("--euler", type=int, default=N_EULER, help="Euler ODE steps")
    args = ap.parse_args()
    B_PER_GPU, T_SRC, T_REF, N_EULER = args.batch, args.frames, args.ref_frames, args.euler

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from serenade_amd import ops
    from serenade_amd.parallel import gather_waveforms
    from serenade_amd.utils.synth import synth_inputs

    model, voc, sd, gsd = build_models(dev)
    d = synth_inputs(B_PER_GPU, T_SRC, T_ref=T_REF, seed=1235 + rank)
    g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}

    def step():
        mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"],
                              g["ref_logmel"], g["ref_midi"], g["ref_lft"], n_timesteps=N_EULER, noise=g["z"])
        wave = voc.decode_batch(mel if mel.dim() == 3 else mel.unsqueeze(0))  # inference() squeezes B == 1
        return gather_waveforms(wave, dst=0, uniform=True) if world > 1 else wave

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import serenade_amd
    if args.graphs is not None:
        ops.set_graphs(bool(args.graphs))

    fl = algorithmic_flops(B_PER_GPU, T_SRC, T_REF, N_EULER)

    def timed(precision):
        """W warm-up + K timed steps in the given contraction precision; HIP events around every conv_gemm launch
        (recorded on the launch stream) give the dominant kernel's time inside the timed region."""
        serenade_amd.set_precision(precision)
        for _ in range(args.warmup):
            step()
        sync()
        # HIP events bracket every contraction launch of the FIRST timed step only: an event pair per launch costs
        # ~3.4 us of drained queue (measured: 87.8 ms/step plain, 92.9 with all 739 launches of every step
        # bracketed), which would be charged to `value`; one step gives 739 samples of the kernel.
        prof = []
        t0 = time.perf_counter()
        for i in range(args.steps):
            ops.PROFILE = prof if i == 0 else None
            step()
        ops.PROFILE = None
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        durs = np.array([s.elapsed_time(e) for s, e in prof], dtype=np.float64)  # ms
        n_launch = float(len(durs))  # launches of one step
        gemm_ms = durs.sum()
        achieved = fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16x3" else PEAK_FP32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": None, "kernel": f"srn_conv_gemm<{precision}> implicit-GEMM contraction kernels "
                          f"({'conv_fast / conv_halo / conv_gemm' if precision == 'bf16x3' else 'conv_gemm'}, all tiles)",
                "launches_per_step": n_launch, "avg_launch_us": (durs.mean() * 1e3) if len(durs) else 0.0,
                "algorithmic_gflop_per_launch": fl / max(n_launch, 1) / 1e9,
                "kernel_time_share": gemm_ms / (elapsed / args.steps * 1e3), "event_sampled_steps": 1}
        if precision == "bf16x3":
            roof["mfma_per_product"] = 3
            roof["frac_of_split_peak"] = achieved / (peak / 3.0)
            headline = (B_PER_GPU, T_SRC, T_REF, N_EULER) == (8, 1024, 256, 10)
            if headline:
                # HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this
                # workload (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); recorded, not live:
                # counters cannot be collected inside a timed run.  profiles/r1_e_pmc_traffic.txt
                roof["traffic"] = 193.49e6
                roof["traffic_unit"] = "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_e_pmc_traffic.txt)"
                roof["algorithmic_bytes_per_launch"] = 133.67e6
        return world * B_PER_GPU * T_SRC * args.steps / elapsed, elapsed, roof

    # headline: split-bf16 contraction (3 bf16 MFMA per fp32 product, fp32 accumulate); parity gates identical
    value, elapsed, roof = timed("bf16x3")
    v32, e32, roof32 = (0.0, 0.0, None) if args.skip_fp32 else timed("fp32")
    out = {
        "metric": f"mel frames/sec converted (UNet ODE + vocoder), 80x{T_SRC}",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 operands as split-bf16 (hi+lo, 3 MFMA/product), f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"B={B_PER_GPU}/GPU x T={T_SRC} source frames (80-dim mel), T_ref={T_REF} prompt, "
                               f"{N_EULER} Euler steps, UNet ODE + HiFi-GAN (8,5,3,2) on GPU; waveform gather to "
                               f"rank 0 when N>1", "global_batch": world * B_PER_GPU, "x_realtime": value / 100.0},
        "roofline": roof,
        "exact_fp32_mode": {"value": v32, "unit": "frames/s", "ms_per_step": e32 / args.steps * 1e3, "dtype": "f32",
                            "roofline": roof32},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, gsd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
