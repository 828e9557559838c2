#!/usr/bin/env python3
"""Produce the committed evidence behind bench.py's `roofline` object (run on the GPU box, from the repo root):

    python3 tools/profile_round.py r2_a [--modes=fp32,bf16x3] [--no-sq]        (GPU box: collect + aggregate)
    python3 tools/profile_round.py r2_a --aggregate-only                        (anywhere: re-read the collected CSVs)

gpurun only brings gpurun_out/ back, so the raw rocprofv3 output lives in gpurun_out/prof_<tag>/ and the summaries
are (re)built from it into profiles/ by the second form in the build container.

For each contraction mode it profiles THE bench.py command (size sweep and CPU leg off) with rocprofv3:
  1. --kernel-trace --stats                    -> profiles/<tag>_kernel_stats_<mode>.csv
  2. --pmc FETCH_SIZE   and   --pmc WRITE_SIZE  (two separate passes, MI355X_MICROARCH.md HBM section)
  3. --pmc SQ_* (wave cycles, waits, MFMA busy, LDS bank conflicts)  -> profiles/<tag>_pmc_sq_<mode>.txt
and writes profiles/<tag>_pmc_traffic.json, which bench.py reads for `roofline.traffic`:
  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced
  reads at 64 B, so fetched bytes = 2 * FETCH_SIZE * 1024 (the guide's correction); WRITE_SIZE is taken as is.
This script never touches the GPU itself (rocprofv3 children do), and never combines --pmc with trace domains.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL_TAGS = ("conv_f32_kernel", "conv_fast_kernel", "conv_halo_kernel", "conv_strip_kernel", "conv_gemm_kernel", "attn_", "resunit_")
SQ = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
      "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU"]


def bench_cmd(mode, steps, warmup):
    return ["python3", "bench.py", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-sweep", "--no-train", "--no-mixed", "--no-post",
            "--modes", mode]


COLLECT = "--aggregate-only" not in sys.argv


def rocprof(outdir, extra, cmd):
    if not COLLECT:
        return
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir, exist_ok=True)
    full = ["rocprofv3"] + extra + ["--output-format", "csv", "-d", outdir, "--"] + cmd
    print("+", " ".join(full), flush=True)
    env = dict(os.environ, TMPDIR="/tmp")
    with open(os.path.join(outdir, "run.log"), "w") as log:
        rc = subprocess.call(full, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT)
    if rc != 0:
        print(open(os.path.join(outdir, "run.log")).read()[-3000:])
        raise SystemExit(f"rocprofv3 failed ({rc})")


def is_contraction(name):
    return any(t in name for t in KERNEL_TAGS)


def counters(outdir):
    """{counter: (sum over contraction dispatches, n dispatches)}, plus per (kernel, grid) averages"""
    tot = collections.defaultdict(float)
    n = collections.defaultdict(int)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(outdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not is_contraction(name):
                continue
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            tot[c] += v
            n[c] += 1
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")
            short = short[: short.find("(SrnConvParams")] if "(SrnConvParams" in short else short[:90]
            per[(short, r["Grid_Size"])][c].append(v)
    return tot, n, per


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r2"
    modes = "fp32,bf16x6,bf16x3"
    for a in sys.argv[1:]:
        if a.startswith("--modes="):
            modes = a.split("=", 1)[1]
    modes = modes.split(",")
    import bench  # host-side helpers only (build_id); does not initialise the GPU
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    scratch = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    build_file = os.path.join(scratch, "build_id.txt")
    if COLLECT:
        os.makedirs(scratch, exist_ok=True)
        open(build_file, "w").write(bench.build_id())
    record = {"build": open(build_file).read().strip(), "workload": "bench.py default: B=8 x T=1024, T_ref=256, 10 Euler steps + "
              "HiFi-GAN (8,5,3,2); 1 warm-up + 1 timed step per counter pass", "modes": {},
              "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; "
                        "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); contraction kernels only"}
    for mode in modes:
        d = os.path.join(scratch, mode + "_kt")
        rocprof(d, ["--kernel-trace", "--stats"], bench_cmd(mode, 3, 1))
        stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(prof, f"{tag}_kernel_stats_{mode}.csv"))
        logs = [ln for ln in open(os.path.join(d, "run.log")) if ln.startswith("{")]
        if logs:
            open(os.path.join(prof, f"{tag}_bench_under_rocprof_{mode}.json"), "w").write(logs[-1])
        df, dw = os.path.join(scratch, mode + "_pmc_f"), os.path.join(scratch, mode + "_pmc_w")
        rocprof(df, ["--pmc", "FETCH_SIZE"], bench_cmd(mode, 1, 1))
        rocprof(dw, ["--pmc", "WRITE_SIZE"], bench_cmd(mode, 1, 1))
        tf, nf, _ = counters(df)
        tw, nw, _ = counters(dw)
        fetch = 2.0 * tf["FETCH_SIZE"] * 1024.0 / max(nf["FETCH_SIZE"], 1)
        write = tw["WRITE_SIZE"] * 1024.0 / max(nw["WRITE_SIZE"], 1)
        record["modes"][mode] = {"launches": nf["FETCH_SIZE"], "fetch_bytes_per_launch": fetch,
                                 "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write}
        print(mode, record["modes"][mode], flush=True)
        if "--no-sq" not in sys.argv:
            # two passes of four counters: a single pass of eight has exceeded the hardware's slots before
            tot, per = collections.defaultdict(float), collections.defaultdict(lambda: collections.defaultdict(list))
            for half in (0, 1):
                ds = os.path.join(scratch, f"{mode}_pmc_sq{half}")
                rocprof(ds, ["--pmc"] + SQ[4 * half: 4 * half + 4], bench_cmd(mode, 1, 1))
                t_, _, p_ = counters(ds)
                for c, v in t_.items():
                    tot[c] += v
                for key, cv in p_.items():
                    for c, v in cv.items():
                        per[key][c] += v
            with open(os.path.join(prof, f"{tag}_pmc_sq_{mode}.txt"), "w") as f:
                f.write(f"# rocprofv3 --pmc {' '.join(SQ)} -- {' '.join(bench_cmd(mode, 1, 1))}\n")
                f.write(f"# build {record['build']}; contraction kernels only; SQ_* in quad-cycles except "
                        f"SQ_VALU_MFMA_BUSY_CYCLES (cycles)\n")
                wc = max(tot["SQ_WAVE_CYCLES"], 1.0)
                f.write(f"# totals: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {tot['SQ_WAIT_ANY'] / wc:.3f}; "
                        f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {tot['SQ_WAIT_INST_ANY'] / wc:.3f}; "
                        f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {tot['SQ_ACTIVE_INST_ANY'] / wc:.3f}; "
                        f"MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES) = "
                        f"{tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(32.0 * tot['SQ_BUSY_CYCLES'], 1.0):.3f}"
                        f"  [busy cycles summed over 1024 SIMDs / dispatch cycles summed over 32 shader engines]\n")
                f.write("kernel | grid | n | " + " | ".join(SQ) + " | mfma_busy_frac | wait_any_frac\n")
                for key, v in sorted(per.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
                    cnt = max(len(x) for x in v.values())
                    av = {c: sum(v[c]) / max(len(v[c]), 1) for c in SQ}
                    f.write(f"{key[0]:70s} {key[1]:>9s} {cnt:4d} " + " ".join(f"{av[c]:12.4g}" for c in SQ) +
                            f" {av['SQ_VALU_MFMA_BUSY_CYCLES'] / max(32.0 * av['SQ_BUSY_CYCLES'], 1.0):8.3f}"
                            f" {av['SQ_WAIT_ANY'] / max(av['SQ_WAVE_CYCLES'], 1.0):8.3f}\n")
    json.dump(record, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print("wrote", os.path.join(prof, f"{tag}_pmc_traffic.json"))


if __name__ == "__main__":
    main()
