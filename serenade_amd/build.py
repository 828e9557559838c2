"""Build libserenade_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries a sidecar `libserenade_hip.so.id` = hash of the kernel sources + C-ABI header it was built from;
`build()` recompiles when that differs from the sources on disk (file times do not survive a copy to the GPU box),
when `force` is set or when SERENADE_AMD_FORCE_BUILD=1, and says which it did."""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libserenade_hip.so")
SOURCES = ["conv_gemm.hip", "conv_halo.hip", "conv_fast.hip", "conv_f32.hip", "conv_splitk.hip", "conv_strip.hip", "resunit.hip", "resunit_f32.hip", "norm_act.hip", "gst.hip", "features.hip", "train.hip", "gst_train.hip", "tn_gemm.hip", "world.hip", "api.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_id():
    """hash of the kernel sources + C ABI (profiles/ records and bench lines carry the same id)"""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def built_id():
    try:
        return open(LIB + ".id").read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_id() != source_id()


LAST_BUILD = None  # "compiled" | "reused" after build()


def build(force=False, verbose=True):
    global LAST_BUILD
    force = force or os.environ.get("SERENADE_AMD_FORCE_BUILD", "0") == "1"
    if not force and not needs_build():
        LAST_BUILD = "reused"
        if verbose:
            print(f"libserenade_hip.so: reused, build id {built_id()} matches the sources", flush=True)
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    headers = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        headers.update(open(f, "rb").read())
    fresh = []
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src + ".o")
        oid = hashlib.sha1(headers.digest() + open(os.path.join(CSRC, src), "rb").read()).hexdigest()
        objs.append(obj)
        try:  # objects of unchanged sources are kept (build/ is scratch: git- and gpurun-ignored)
            if not force and os.path.exists(obj) and open(obj + ".id").read() == oid:
                continue
        except OSError:
            pass
        fresh.append((obj, oid))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for obj, oid in fresh:
        open(obj + ".id", "w").write(oid)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(LIB + ".id", "w") as f:
        f.write(source_id() + "\n")
    LAST_BUILD = "compiled"
    if verbose:
        print(f"libserenade_hip.so: compiled {len(fresh)} of {len(SOURCES)} sources, build id {source_id()}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
