#!/usr/bin/env python3
"""Developer tool: rewrite the numbers block (between the `<!-- numbers:begin -->` / `<!-- numbers:end -->` markers) of
DESIGN.md and README.md from ONE bench.py JSON line, so every figure in the two documents comes from the same run.

    python3 tools/update_numbers.py <file holding the bench line> ["label of the run"]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
    label = sys.argv[2] if len(sys.argv) > 2 else "bench.py defaults"
    f = lambda v, n=0: f"{v:,.{n}f}".replace(",", " ")
    r = d["roofline"]
    x6, x3, mix = d["fp32_emulated_bf16x6_mode"], d["split_bf16_mode"], d["fp32_with_bf16x3_attention_mode"]
    t = d["train_step"]
    cb = d["cpu_baseline"]
    sw = {(s["B"], s["T"], s["euler_steps"]): s for s in d["sweep"]}
    ms = lambda k, m="fp32": f(sw[k][m]["ms_per_batch"], 1 if sw[k][m]["ms_per_batch"] < 100 else 0)
    an = d["analysis_stage"]
    tr = r.get("traffic")
    lines = [
        f"Round-4 numbers (1× MI355X, {label}, build `{d['config'].get('build', '?')}`; boxes differ by ±4 %):",
        "",
        "| B = 8 × T = 1024, 10 Euler steps + HiFi-GAN | frames/s | ms / step | contraction TFLOP/s | of roof |",
        "|---|---|---|---|---|",
        f"| **exact fp32 (headline)** | **{f(d['value'])}** ({f(d['value'] / 100)}× real time) | {f(d['ms_per_step'], 1)} | "
        f"{f(r['achieved'], 1)} (kernel time), {f(r['achieved_of_step'], 1)} (whole step) | **{r['frac']:.3f}** / "
        f"{r['frac_of_step']:.3f} of 157.3 |",
        f"| fp32 convs / linears + bf16x3 attention | {f(mix['value'])} | {f(mix['ms_per_step'], 1)} | | secondary key |",
        f"| bf16x6 (fp32-faithful) | {f(x6['value'])} | {f(x6['ms_per_step'], 1)} | {f(x6['roofline']['achieved'], 1)} | "
        f"{x6['roofline']['frac_of_split_peak']:.3f} of 416.7 |",
        f"| bf16x3 | {f(x3['value'])} | {f(x3['ms_per_step'], 1)} | {f(x3['roofline']['achieved'], 1)} | "
        f"{x3['roofline']['frac_of_split_peak']:.3f} of 833.3 |",
        f"| CPU oracle on the host, {cb['cores']} threads / 1 thread (same utterance, T = 1024) | {f(cb['value'])} / "
        f"{f(cb['one_thread']['value'], 1)} | | | |",
        "",
        f"Round 3's headline was 57 705 frames/s = 0.726 (kernel time); this round's builds measured 59.7–62.5 k frames/s "
        f"(0.76–0.79) across the boxes they ran on.  HBM-side traffic of the contraction kernels: "
        + (f"{tr / 1e6:.0f} MB per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic "
           f"({tr / r['algorithmic_bytes_per_launch']:.2f}×; `{r['traffic_record']['source']}`, measured on build "
           f"`{r['traffic_record']['measured_on_build']}`)." if tr else "not recorded for this build."),
        "",
        "| other sizes, exact fp32 (ms per batch) | |",
        "|---|---|",
        f"| B = 1 × T = 256, 10 steps (the reference CLI's operating point, C1) | {ms((1, 256, 10))} |",
        f"| B = 8 × T = 256 / B = 1 × T = 1024 / B = 8 × T = 1024 with 20 steps (C3) | {ms((8, 256, 10))} / {ms((1, 1024, 10))} / "
        f"{ms((8, 1024, 20))} |",
        f"| B = 8 × T = 4096, 10 steps (C5's shape; 709 in round 3) | {ms((8, 4096, 10))}; with bf16x3 attention "
        f"{ms((8, 4096, 10), 'fp32_with_bf16x3_attention')} |",
        f"| analysis between HiFi-GAN and SiFiGAN, 8 × 10.24 s = {an['frames']} frames | {an['analyzer_total']['ms']:.1f} ms; "
        f"SiFiGAN generator behind it {an['sifigan_generator']['ms']:.1f} ms |",
        f"| whole-model training step, B = 4 × L = 1024: eager / captured as a hipGraph; B = 16 | {f(t['ms_per_step'], 1)} / "
        f"{f(t['captured_as_hipgraph']['ms_per_step'], 1)} ms ({f(t['captured_as_hipgraph']['tflops'], 1)} TFLOP/s); "
        f"{f(t['at_batch_16']['ms_per_step'], 1)} ms |",
    ]
    block = "<!-- numbers:begin -->\n" + "\n".join(lines) + "\n<!-- numbers:end -->"
    for name in ("DESIGN.md", "README.md"):
        p = os.path.join(ROOT, name)
        s = open(p).read()
        if "<!-- numbers:begin -->" in s:
            s = re.sub(r"<!-- numbers:begin -->.*?<!-- numbers:end -->", lambda m: block, s, flags=re.S)
        else:
            s = s.replace("@@NUMBERS@@", block)
        open(p, "w").write(s)
    print(block)


if __name__ == "__main__":
    main()
