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

The top-level `value` / `ms_per_step` / `dtype` / `roofline` are measured in the REFERENCE's precision: exact fp32
contraction on v_mfma_f32_32x32x2_f32.  The faster split-bf16 product mode (operands as bf16 hi+lo, 3 MFMA per
product: inside the north-star tolerances but narrower than fp32) and the fp32-faithful bf16x6 emulation (exact
three-way operand split, 6 MFMA per product) are timed in the same run and reported under `split_bf16_mode` /
`fp32_emulated_bf16x6_mode`; the north-star's other sizes under `sweep`; the CPU oracle under `cpu_baseline`.
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
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
DTYPE = {"fp32": "f32", "bf16x3": "f32 operands as split-bf16 (hi+lo, 3 MFMA/product), f32 accumulate",
         "bf16x6": "f32 emulated on bf16 MFMA: exact hi+mid+lo operand split (24 significand bits), 6 MFMA/product, "
                   "f32 accumulate (dropped terms <= 2^-26 per product); the fused HiFi-GAN residual units (C = 32 / 64) "
                   "and the generic / halo / strip conv kernels have no x6 variant and run exact f32 MFMA in this mode"}
MODE_KEY = {"fp32": "exact_fp32_mode", "bf16x3": "split_bf16_mode", "bf16x6": "fp32_emulated_bf16x6_mode"}


def algorithmic_flops(B, T, T_ref, n):
    """SURVEY.md section 8(d) per-utterance work: estimator 81.3e6*L + 24576*L^2 per call, encoder 11.32e6 per
    frame, HiFi-GAN 499e6 per source frame (all of it runs in the conv_gemm kernel; GST's 0.8 GFLOP does not)."""
    L = T + T_ref
    return B * (n * (81.3e6 * L + 24576.0 * L * L) + 11.32e6 * (T + T_ref) + 499e6 * T)


def train_step_bench(dev, sd, B=4, L=1024, steps=3):
    """SURVEY 8 f4, a secondary line: one training step of the whole model as the reference's trainer runs it
    (trainers/ssc.py:57-96: Serenade.forward -> cfm_loss + prior_loss -> backward -> clip_grad_norm_(1.0) -> AdamW
    lr 8e-4; dropout 0.05) at its per-GPU batch (conf/serenade.yaml:52 batch_size 4) on L = 1024 frames, exact-fp32
    contraction.  FLOPs = 3 x the forward of estimator + encoder (the GST's 0.8 GFLOP is not counted)."""
    from serenade_amd import training
    model = training.TrainSerenade(sd, dev, dropout=0.05)
    opt = training.AdamW(model)
    sync = training.GradSync(model)
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)  # plans of the inference benchmark stay resident
    g = torch.Generator().manual_seed(4321)
    lens = torch.tensor([L - 37 * i for i in range(B)]).to(dev)
    x, logmel = torch.randn(B, L, 768, generator=g).to(dev), torch.randn(B, L, 80, generator=g).to(dev)
    midi, lft = torch.randn(B, L, 1, generator=g).to(dev), torch.randn(B, L, 1, generator=g).to(dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    phases = {"forward_ms": 0.0, "backward_ms": 0.0, "optimizer_ms": 0.0}
    loss0 = loss = None
    for it in range(steps + 1):  # one warm-up
        e = [ev() for _ in range(4)]
        e[0].record()
        ret = model(x, lens, logmel, midi, lft)
        loss = ret["cfm_loss"] + ret["prior_loss"]
        e[1].record()
        model.backward(loss, sync)  # one rank: autograd.grad + one multi-tensor copy; several: hooks + bucketed all-reduce
        e[2].record()
        opt.step()
        e[3].record()
        torch.cuda.synchronize()
        if it == 0:
            loss0 = float(loss.detach())
            t0 = time.perf_counter()
        else:
            for k, (a, b) in zip(phases, ((0, 1), (1, 2), (2, 3))):
                phases[k] += e[a].elapsed_time(e[b]) / steps
    dt = (time.perf_counter() - t0) / steps
    fl = 3.0 * B * (81.3e6 * L + 24576.0 * L * L + 11.32e6 * L)
    # the same step captured as one hipGraph (training.GraphedStep): no host-side launch overhead
    del opt, sync
    try:
        model = training.TrainSerenade(sd, dev, dropout=0.05)
        gopt = training.AdamW(model)
        gstep = training.GraphedStep(model, gopt, B, L)
        for _ in range(2):
            gstep(x, lens, logmel, midi, lft)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(steps):
            cfm, prior, _ = gstep(x, lens, logmel, midi, lft)
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - tg) / steps
        graphed = {"ms_per_step": dtg * 1e3, "frames_per_s": B * L / dtg, "tflops": fl / dtg / 1e12,
                   "loss_last": float(cfm + prior)}
    except Exception as e:  # noqa: BLE001  (capture depends on library state; the eager numbers stand on their own)
        graphed = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"workload": f"whole-model training step (encoder + GST + estimator, cfm + prior loss), B={B} x L={L} "
                        f"(ragged lengths), exact fp32, dropout 0.05, clip 1.0, AdamW lr 8e-4",
            "ms_per_step": dt * 1e3, "frames_per_s": B * L / dt, "tflops": fl / dt / 1e12,
            "dropout_timed_at": 0.05,
            "dropout_validated_at": 0.0,  # the reference's mask draws cannot be reproduced: gradient parity runs use p = 0
            "parameters": int(sum(v.numel() for v in model.params.values())), **phases,
            "loss_first": loss0, "loss_last": float(loss.detach()), "captured_as_hipgraph": graphed,
            "peak_hbm_gib": (torch.cuda.max_memory_allocated(dev) - base) / 2**30}


def build_id():
    """hash of the kernel sources + C ABI the loaded library was built from (profiles/ records carry the same id)"""
    from serenade_amd import build
    return build.source_id()


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


# ---------------------------------------------------------------------------------------------------- CPU baseline
def host_cpu():
    """(model name, physical cores of the machine, CPUs this process may run on)"""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), usable


def cpu_baseline(sd, gsd):
    """The CPU oracle (a port of the reference's CPU path, pinned to the reference's golden vectors) timed on this
    box's host cores, fp32 torch CPU, same synthetic inputs, bounded samples (SURVEY section 8d / BASELINE.md 4):
      * all usable physical cores: ONE utterance of the headline workload (T=1024, T_ref=256, 10 Euler steps +
        HiFi-GAN) -- 1 warm-up on a short clip, median of 3;
      * 1 thread (the recipe's OMP_NUM_THREADS=1, egs/gtsinger/ssc1/path.sh:16): the SAME utterance (T=1024), one
        timed run (~17 s), so the two legs can be set side by side; the C1 shape (T=256, T_ref=256) on one
        thread as before, median of 3."""
    from oracle import serenade_oracle as O
    from serenade_amd.utils.synth import HIFIGAN_PARAMS, synth_inputs
    t1 = torch.ones(80)
    ident = {"mean": 0 * t1, "scale": t1}

    def run(d):
        with torch.no_grad():
            t0 = time.perf_counter()
            mel = O.serenade_inference(sd, d["x"], d["lengths"], d["midi"], d["lft"], d["ref_x"], d["ref_lengths"],
                                       d["ref_logmel"], d["ref_midi"], d["ref_lft"], d["z"], n_timesteps=N_EULER)
            O.vocoder_decode(gsd, mel, HIFIGAN_PARAMS, ident, ident)
            return time.perf_counter() - t0

    model, phys, usable = host_cpu()
    n_all = max(1, min(phys, usable))
    saved = torch.get_num_threads()
    warm = synth_inputs(1, 64, T_ref=32, seed=7)
    legs = {}
    try:
        for name, threads, T, reps in (("all_cores", n_all, 1024, 3), ("one_thread", 1, 1024, 1),
                                       ("one_thread_c1_shape", 1, 256, 3)):
            torch.set_num_threads(threads)
            d = synth_inputs(1, T, T_ref=256, seed=1235)
            run(warm)
            ts = [run(d) for _ in range(reps)]
            med = statistics.median(ts)
            legs[name] = {"value": T / med, "unit": "frames/s", "threads": threads, "median_s": med,
                          "runs_s": [round(t, 3) for t in ts],
                          "sample": f"1 utterance T={T}, T_ref=256, {N_EULER} Euler steps + HiFi-GAN (8,5,3,2)"}
    finally:
        torch.set_num_threads(saved)
    a = legs["all_cores"]
    return {"value": a["value"], "unit": "frames/s", "cores": a["threads"], "kind": "port",
            "sample": a["sample"] + f"; fp32 torch CPU, 1 warm-up, median of 3 ({a['median_s']:.1f} s)",
            "cpu_model": model, "physical_cores": phys, "usable_cpus": usable, "one_thread": legs["one_thread"],
            "one_thread_c1_shape": legs["one_thread_c1_shape"]}


def cpu_train_baseline(sd, B=1, L=256):
    """the training step's CPU side-by-side: Serenade.forward in train mode + backward by torch autograd through the
    oracle on the usable host cores (no optimizer), one warm-up on a short clip, median of 3"""
    from oracle import serenade_oracle as O
    _, phys, usable = host_cpu()
    threads = max(1, min(phys, usable))
    saved = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        def run(T):
            g = torch.Generator().manual_seed(5)
            w = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))
                     else v) for k, v in sd.items()}
            x, mel = torch.randn(B, T, 768, generator=g), torch.randn(B, T, 80, generator=g)
            midi, lft = torch.randn(B, T, 1, generator=g), torch.randn(B, T, 1, generator=g)
            t, z = torch.rand(B, 1, 1, generator=g), torch.randn(B, 80, T, generator=g)
            t0 = time.perf_counter()
            ret = O.serenade_forward(w, x, [T] * B, mel, midi, lft, 0.3, T // 4, t, z, bn_training=True)
            (ret["cfm_loss"] + ret["prior_loss"]).backward()
            return time.perf_counter() - t0
        run(64)
        ts = [run(L) for _ in range(3)]
    finally:
        torch.set_num_threads(saved)
    med = statistics.median(ts)
    return {"value": B * L / med, "unit": "frames/s", "cores": threads, "kind": "port", "median_s": med,
            "sample": f"B={B} x L={L} forward + backward through the oracle (torch autograd, fp32), median of 3",
            "workload_differs_from_gpu_line": f"CPU leg: B={B} x L={L}, no clip / AdamW; GPU line: B=4 x L=1024 with "
                                              "them -- context only, do not divide the two"}


# ---------------------------------------------------------------------------------------------------- traffic record
def recorded_traffic(precision):
    """HBM-side bytes per contraction launch from the newest committed PMC record (tools/profile_round.py writes
    profiles/r*_pmc_traffic.json from two separate rocprofv3 --pmc passes over this command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside a timed run, so the number is
    read from the record, together with the build it was measured on."""
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not recs:
        return None
    try:
        rec = json.load(open(recs[-1]))
        m = rec["modes"][precision]
    except (OSError, ValueError, KeyError):
        return None
    return {"bytes_per_launch": m["traffic_bytes_per_launch"], "fetch_bytes_per_launch": m["fetch_bytes_per_launch"],
            "write_bytes_per_launch": m["write_bytes_per_launch"], "launches": m["launches"],
            "source": os.path.relpath(recs[-1], ROOT), "measured_on_build": rec.get("build"),
            "workload": rec.get("workload")}


def op_bytes(kw):
    """operand bytes one contraction launch must move once: A + B read, output written, residuals read"""
    if "dilation" in kw:  # fused HiFi-GAN residual unit: x in, y out, two weight tensors (+ the stage sum)
        t = kw["n_batch"] * kw["T"] * kw["C"] * 4
        return t * (2 + (kw.get("res2") is not None)) + 2 * kw["k"] * kw["C"] * kw["C"] * 4
    Z = kw["n_batch"] * kw.get("n_head", 1)
    taps = len(kw.get("taps", (0,)))
    n_out = kw.get("N_out", 0) or kw["N"]
    rd = Z * kw["T_in"] * kw["C_in"] * 4
    per_z = bool(kw.get("w_bs", 0) or kw.get("w_hs", 0))
    rd += (Z if per_z else 1) * kw["N"] * taps * kw["C_in"] * 4
    out_b = Z * kw["T_out"] * n_out * 4
    rd += out_b * ((kw.get("res") is not None) + (kw.get("res2") is not None))
    return rd + out_b


def main():
    global B_PER_GPU, T_SRC, T_REF, N_EULER
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the north-star size sweep (profiling runs)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the fp32 + bf16x3-attention secondary key (profiling runs)")
    ap.add_argument("--no-train", action="store_true", help="skip the estimator training-step line (SURVEY 8 f4)")
    ap.add_argument("--modes", default="fp32,bf16x6,bf16x3", help="contraction modes to time; the first is the "
                                                                  "headline (profiling runs pass one)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the "
                    "waveform gather even with one rank (single-GPU rehearsal of the N>1 path)")
    ap.add_argument("--no-post", action="store_true", help="skip the analysis-stage (f1) secondary key")
    ap.add_argument("--graphs", type=int, default=None, help="1/0: replay the plans as hipGraphs (default: library default)")
    # the north-star's other sizes (T in {256, 1024, 4096}, 20 Euler steps); the defaults are the headline workload
    ap.add_argument("--frames", type=int, default=T_SRC, help="source mel frames per utterance")
    ap.add_argument("--ref-frames", type=int, default=T_REF, help="prompt frames per utterance")
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="utterances per GPU")
    ap.add_argument("--euler", type=int, default=N_EULER, help="Euler ODE steps")
    args = ap.parse_args()
    B_PER_GPU, T_SRC, T_REF, N_EULER = args.batch, args.frames, args.ref_frames, args.euler
    modes = [m for m in args.modes.split(",") if m]
    assert modes and all(m in DTYPE for m in modes), f"--modes takes {sorted(DTYPE)}"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:  # before any other GPU call of this process
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import serenade_amd
    from serenade_amd import ops
    from serenade_amd.parallel import gather_waveforms
    from serenade_amd.utils.synth import synth_inputs

    model, voc, sd, gsd = build_models(dev)
    if args.graphs is not None:
        ops.set_graphs(bool(args.graphs))
    gather_ev = []  # (start, end) HIP events around the waveform gather of the timed steps

    def make_step(B, T, T_ref, n, seed):
        d = synth_inputs(B, T, T_ref=T_ref, seed=seed)
        g = {k: (v.to(dev) if v.is_floating_point() else v) for k, v in d.items()}

        def step(timed=False):
            mel = model.inference(g["x"], g["lengths"], g["midi"], g["lft"], g["ref_x"], g["ref_lengths"],
                                  g["ref_logmel"], g["ref_midi"], g["ref_lft"], n_timesteps=n, noise=g["z"])
            wave = voc.decode_batch(mel if mel.dim() == 3 else mel.unsqueeze(0))  # inference() squeezes B == 1
            if not use_dist:
                return wave
            if timed:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = gather_waveforms(wave, dst=0, uniform=True, always_collective=True)
                e.record()
                gather_ev.append((s, e))
                return out
            return gather_waveforms(wave, dst=0, uniform=True, always_collective=True)

        return step

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step = make_step(B_PER_GPU, T_SRC, T_REF, N_EULER, 1235 + rank)
    fl = algorithmic_flops(B_PER_GPU, T_SRC, T_REF, N_EULER)

    def timed(precision):
        """W warm-up + K timed steps in the given contraction precision; HIP events around every conv_gemm launch
        (recorded on the launch stream) give the dominant kernel's time inside the timed region."""
        serenade_amd.set_precision(precision)
        for _ in range(args.warmup):
            step()
        sync()
        # HIP events bracket every contraction launch of the FIRST timed step only: an event pair per launch costs
        # ~3.4 us of drained queue (739 launches: ~2.5 ms), which would otherwise be charged to `value` on every
        # step; one step gives 739 samples of the kernel.
        prof = []
        del gather_ev[:]
        t0 = time.perf_counter()
        for i in range(args.steps):
            ops.PROFILE = prof if i == 0 else None
            step(timed=True)
        ops.PROFILE = None
        sync()
        elapsed = local_elapsed = time.perf_counter() - t0
        extra = {}
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            elapsed = max(float(x.item()) for x in every)
            g_ms = [s.elapsed_time(e) for s, e in gather_ev]
            extra = {"per_rank_ms_per_step": [float(x.item()) / args.steps * 1e3 for x in every],
                     "rank0_gather_ms_per_step": sum(g_ms) / max(len(g_ms), 1),
                     "world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                     "gather_bytes_per_rank": B_PER_GPU * T_SRC * 240 * 4}
        durs = np.array([s.elapsed_time(e) for s, e, _ in prof], dtype=np.float64)  # ms
        n_launch = float(len(durs))  # launches of one step
        gemm_ms = durs.sum()
        alg_bytes = float(sum(op_bytes(op.kw) for _, _, op in prof))
        achieved = fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        tr = recorded_traffic(precision)
        same_workload = (B_PER_GPU, T_SRC, T_REF, N_EULER) == (8, 1024, 256, 10)
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": tr["bytes_per_launch"] if (tr and same_workload) else None,
                # the same algorithmic FLOPs over the WHOLE driver-timed step (small kernels, launch gaps, host included)
                "frac_of_step": fl / (elapsed / args.steps) / 1e12 / peak,
                "achieved_of_step": fl / (elapsed / args.steps) / 1e12,
                "kernel": f"srn_conv_gemm<{precision}> implicit-GEMM contraction kernels (conv_f32 / conv_fast / conv_halo "
                          f"/ conv_strip / conv_gemm, all tiles) + srn_hifigan_resunit",
                "launches_per_step": n_launch, "avg_launch_us": (durs.mean() * 1e3) if len(durs) else 0.0,
                "algorithmic_gflop_per_launch": fl / max(n_launch, 1) / 1e9,
                "algorithmic_bytes_per_launch": alg_bytes / max(n_launch, 1),
                "kernel_time_share": gemm_ms / (local_elapsed / args.steps * 1e3), "event_sampled_steps": 1}
        if tr and same_workload:
            roof["traffic_record"] = {k: tr[k] for k in ("source", "measured_on_build", "fetch_bytes_per_launch",
                                                         "write_bytes_per_launch", "launches")}
            roof["traffic_record"]["current_build"] = build_id()
            roof["traffic_unit"] = "bytes/launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
        if precision != "fp32":
            per = 3 if precision == "bf16x3" else 6
            roof["mfma_per_product"] = per
            roof["frac_of_split_peak"] = achieved / (peak / per)
        value = world * B_PER_GPU * T_SRC * args.steps / elapsed
        return {"value": value, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                "dtype": DTYPE[precision], "x_realtime": value / 100.0, "roofline": roof, **extra}

    results = {m: timed(m) for m in modes}
    head = results[modes[0]]

    def mixed_attention():
        """a secondary key: exact-fp32 convs / linears with Q K^T and P V on the bf16 matrix cores (split-bf16, fp32
        accumulate; serenade_amd.set_attention_precision) -- BASELINE configs[4]'s "MFMA attention" on the headline
        workload.  Same warm-up / step counts; no per-launch events."""
        serenade_amd.set_precision("fp32")
        serenade_amd.set_attention_precision("bf16x3")
        try:
            for _ in range(args.warmup):
                step()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            dt = (time.perf_counter() - t0) / args.steps
        finally:
            serenade_amd.set_attention_precision(None)
        return {"value": world * B_PER_GPU * T_SRC / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                "dtype": "f32 convs / linears; attention contractions on split-bf16 operands (3 MFMA/product), f32 "
                         "accumulate", "tflops": fl / dt / 1e12,
                "accuracy": "mel 6e-7 relative / waveform 3e-6 absolute against the all-fp32 run at T = 4096 "
                            "(tools/c5bench.py); gates held in tests/test_hip_configs.py::test_c5_mfma_attention_option"}

    def sweep():
        """the north-star's sizes on this GPU: frames/s (source frames only; the UNet also carries the 256-frame
        prompt), 1 warm-up + 2 timed steps each, no per-launch events"""
        rows = []
        for T in (256, 1024, 4096):
            for n in (10, 20):
                for B in (1, 8):
                    st = make_step(B, T, 256, n, 1235)
                    row = {"B": B, "T": T, "T_ref": 256, "euler_steps": n}
                    for m in modes:
                        serenade_amd.set_precision(m)
                        st()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(2):
                            st()
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) / 2
                        row[m] = {"frames_per_s": B * T / dt, "ms_per_batch": dt * 1e3,
                                  "tflops": algorithmic_flops(B, T, 256, n) / dt / 1e12}
                    if T == 4096 and n == 10:
                        # BASELINE configs[4]'s "MFMA attention": exact-fp32 convs / linears, Q K^T and P V on the bf16
                        # matrix cores (serenade_amd.set_attention_precision); gates held in tests/test_hip_configs.py
                        serenade_amd.set_precision("fp32")
                        for attn in ("bf16x6", "bf16x3"):
                            serenade_amd.set_attention_precision(attn)
                            try:
                                st()
                                torch.cuda.synchronize()
                                t0 = time.perf_counter()
                                for _ in range(2):
                                    st()
                                torch.cuda.synchronize()
                                dt = (time.perf_counter() - t0) / 2
                            finally:
                                serenade_amd.set_attention_precision(None)
                            row[f"fp32_with_{attn}_attention"] = {
                                "frames_per_s": B * T / dt, "ms_per_batch": dt * 1e3,
                                "tflops": algorithmic_flops(B, T, 256, n) / dt / 1e12}
                    rows.append(row)
        return rows

    out = {
        "metric": f"mel frames/sec converted (UNet ODE + vocoder), 80x{T_SRC}",
        "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
        "config": {"workload": f"B={B_PER_GPU}/GPU x T={T_SRC} source frames (80-dim mel), T_ref={T_REF} prompt, "
                               f"{N_EULER} Euler steps, UNet ODE + HiFi-GAN (8,5,3,2) on GPU; waveform gather to "
                               f"rank 0 when N>1", "global_batch": world * B_PER_GPU,
                   "x_realtime": head["x_realtime"], "precision_vs_reference": "same (fp32)" if modes[0] == "fp32"
                   else "narrower than the reference's fp32 (split-bf16 operands)", "build": build_id()},
        "roofline": head["roofline"],
    }
    for k in ("per_rank_ms_per_step", "rank0_gather_ms_per_step", "world_size_seen", "backend",
              "gather_bytes_per_rank"):
        if k in head:
            out.setdefault("multi_gpu", {})[k] = head[k]
    for m in modes[1:]:
        out[MODE_KEY[m]] = results[m]
    if "fp32" in modes and world == 1 and not args.no_mixed:
        out["fp32_with_bf16x3_attention_mode"] = mixed_attention()
    if rank == 0 and world == 1 and not args.no_train:
        # a secondary line: it must never cost the headline its JSON
        try:
            serenade_amd.set_precision(modes[0])
            out["train_step"] = train_step_bench(dev, sd)
            # the same step at B = 16: at the reference's B = 4 the step is bound by its ~2900 launches
            big = train_step_bench(dev, sd, B=16)
            out["train_step"]["at_batch_16"] = {k: big[k] for k in ("ms_per_step", "frames_per_s", "tflops", "forward_ms",
                                                                    "backward_ms", "optimizer_ms",
                                                                    "captured_as_hipgraph", "peak_hbm_gib")}
        except Exception as e:  # noqa: BLE001
            out["train_step"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_post:
        # SURVEY 8 f1, a secondary key: the analysis front-end between HiFi-GAN and SiFiGAN (CheapTrick, mel-cepstrum, D4C,
        # F0 contours, excitation; fp64, one workgroup per 5 ms frame) on the waveforms of the same batch size, and the
        # SiFiGAN generator behind it -- tools/worldbench.py's numbers
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location(
                "worldbench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "worldbench.py"))
            wb = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(wb)
            serenade_amd.set_precision(modes[0])
            out["analysis_stage"] = wb.run(B_PER_GPU, T_SRC, dev)
        except Exception as e:  # noqa: BLE001  (never costs the headline its JSON)
            out["analysis_stage"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_sweep:
        out["sweep"] = sweep()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, gsd)
        if "train_step" in out and "error" not in out["train_step"]:
            try:
                out["train_step"]["cpu_baseline"] = cpu_train_baseline(sd)
            except Exception as e:  # noqa: BLE001
                out["train_step"]["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
