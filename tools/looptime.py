#!/usr/bin/env python3
"""Developer tool (GPU, SRN_DBG_TIMING build of the library): per-wave cycle split of conv_fast's main loop into
issue (loads + cursor bumps) / MFMA+staging block / barrier wait, for one plain GEMM.

    SERENADE_AMD_LIB=build_dbg/lib_TIMING.so python tools/looptime.py [M N K]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import _lib, ops  # noqa: E402


def main():
    a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [10240, 2048, 2048]
    M, N, K = a
    dev = torch.device("cuda:0")
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    op = ops.ConvOp(in0=x, w=w, out=out, n_batch=1, T_in=M, T_out=M, C_in=K, N=N, ld_in0=K, ldw=K, ld_out=N,
                    precision=_lib.PREC_BF16X3, tile=1)
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = (ctypes.c_ulonglong * (1024 * 16))()
    assert lib.srn_dbg_timing(buf, 1024 * 16) == 0
    d = np.array(buf, dtype=np.float64).reshape(1024, 4, 4)
    iters = d[..., 3].max()
    per = d[..., :3] / (2 * iters)  # cycles per k-step
    print(f"M={M} N={N} K={K}: {int(iters) * 2} timed k-steps per block; cycles per k-step per wave "
          f"(mean over {per.shape[0]} blocks x 4 waves)")
    for name, i in (("issue (loads, bumps)", 0), ("MFMA + staging block", 1), ("barrier wait", 2)):
        v = per[..., i]
        print(f"  {name:22s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}")
    print(f"  total                  mean {per.sum(-1).mean():8.0f}   (24 MFMA x 32 cycles = 768)")


if __name__ == "__main__":
    main()
