#!/usr/bin/env python3
"""Developer tool: cut ONE training step out of a `rocprofv3 --kernel-trace --output-format csv` trace of
tools/trainbench.py (between two AdamW launches) and print it by kernel category and by kernel.

    python tools/trainstep_breakdown.py <..._kernel_trace.csv> [out.csv]
"""
import collections
import csv
import sys


def cat(n):
    if "conv_f32" in n or "conv_fast" in n or "conv_gemm_kernel" in n or "splitk" in n or "conv_halo" in n:
        return "own contraction kernels (srn_conv_gemm)"
    if "tn_gemm" in n or "tn_lean" in n or "tn_reduce" in n:
        return "own time-contraction kernels (srn_tn_gemm: wgrad / dK / dV)"
    if n.startswith("Cijk") or "rocblas" in n:
        return "rocBLAS"
    if "anonymous namespace" in n and "at::native" not in n:
        return "own other kernels (train.hip, norm_act.hip)"
    low = n.lower()
    if "miopen" in low or "naive_conv" in n or "batchnorm" in low or "gfx9" in n or "_ZN2ck" in n or "Op2dTensor" in n \
            or "Rnn" in n or "OpTensor" in n:
        return "MIOpen / CK (GST convs, BatchNorm, GRU)"
    return "torch elementwise / reduce / copy / fill"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    lines = ["category,launches,ms,percent_of_kernel_time"]
    for label, key in (("one eager step (zero_grad .. AdamW)", lambda n: "adamw_kernel" in n and "dyn" not in n),
                       ("one captured step (hipGraph replay)", lambda n: "adamw_dyn_kernel" in n)):
        marks = [i for i, n in enumerate(names) if key(n)]
        if len(marks) < 2:
            continue
        seg = rows[marks[-2] + 1: marks[-1] + 1]
        dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot = sum(dur(r) for r in seg)
        span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
        c, k = collections.Counter(), collections.Counter()
        ck, kk = collections.Counter(), collections.Counter()
        for r in seg:
            c[cat(r["Kernel_Name"])] += dur(r)
            k[cat(r["Kernel_Name"])] += 1
            ck[r["Kernel_Name"][:80]] += dur(r)
            kk[r["Kernel_Name"][:80]] += 1
        lines.append(f"# {label}: {len(seg)} launches, kernel time {tot / 1e6:.2f} ms, first start to last end "
                     f"{span / 1e6:.2f} ms (under rocprofv3 --kernel-trace)")
        for n, v in c.most_common():
            lines.append(f"{n},{k[n]},{v / 1e6:.3f},{100 * v / tot:.1f}")
        print(lines[-len(c) - 1])
        for n, v in ck.most_common(40):
            print(f"   {v / 1e3:8.1f} us {kk[n]:4d}  {n}")
    print("\n".join(lines))
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
