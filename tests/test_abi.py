"""CPU checks of the C-ABI boundary: the library loads, exports every symbol include/serenade_hip.h declares,
and the ctypes mirror of SrnConvParams has the C layout (checked with a gcc-compiled probe)."""
import ctypes
import os
import re
import subprocess


import pytest

from serenade_amd import _lib, build, training

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "serenade_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(srn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)  # no-op when up to date; hipcc cross-compiles without a GPU
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in serenade_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_error_reporting_without_gpu(lib):
    assert lib.srn_abi_version() == 3
    assert lib.srn_conv_gemm(None, None) == -1
    assert b"null params" in lib.srn_last_error()
    p = _lib.SrnConvParams()
    assert lib.srn_conv_gemm(ctypes.byref(p), None) == -1  # argument validation happens before any launch
    assert b"null in0" in lib.srn_last_error()


@pytest.mark.parametrize("name", ["SrnConvParams", "SrnResUnitParams", "SrnCopyList", "SrnWorldParams", "SrnExcitationParams", "SrnTnGemmParams", "SrnTransposeList"])
def test_struct_layout_matches_c(tmp_path, name):
    cls = getattr(_lib, name)
    fields = [f[0] for f in cls._fields_]
    probe = tmp_path / "probe.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){',
             f'printf("%zu\\n", sizeof({name}));']
    for f in fields:
        lines.append(f'printf("%zu\\n", offsetof({name}, {f}));')
    lines.append("return 0;}")
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-o", str(exe), str(probe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(cls)
    for f, off in zip(fields, out[1:]):
        assert getattr(cls, f).offset == off, f


def test_partial_sum_chunk_sizes_match_the_host_code():
    """training.py sizes its partial-sum buffers as ceil(T / 32) rows per chunk; the library says the same (pure host
    functions, no GPU needed)"""
    from serenade_amd import _lib
    h = _lib.lib()
    for T in (1, 31, 32, 33, 1000, 4352):
        R = training.NORM_BWD_ROWS  # the host allocates the partial sums by this constant
        assert h.srn_rowln_chunks(T) == (T + R - 1) // R == h.srn_gn_chunks(T)
    assert 1 <= h.srn_sumsq_blocks(10) <= h.srn_sumsq_blocks(84_287_728) <= 1024
    for R in (1, 32, 33, 4096):
        assert h.srn_colsum_chunks(R) == (R + 31) // 32
