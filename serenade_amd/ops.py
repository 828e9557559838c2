"""Thin Python wrappers over the C ABI (include/serenade_hip.h).

Tensors are torch CUDA fp32 tensors used as device memory only; every wrapper passes raw
pointers, sizes and the current HIP stream.  All activations are channels-last (B, T, C).
Nothing here computes on the host and nothing falls back to torch ops or to ``oracle/``.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_LEAKY, ACT_MISH, ACT_NONE, ACT_SILU, POST_DIV, POST_LEAKY, POST_NONE, POST_RELU,  # noqa: F401
                   POST_TANH,
                   RES_ADD, RES_AXPY, RES_NONE, SrnConvParams, SrnResUnitParams, check)


DEFAULT_PRECISION = _lib.PREC_FP32  # contraction arithmetic of ops built without an explicit precision
ATTENTION_PRECISION = None  # Q K^T / P V of the estimator's self-attention; None: DEFAULT_PRECISION


def attention_precision():
    return DEFAULT_PRECISION if ATTENTION_PRECISION is None else ATTENTION_PRECISION
NO_HALO = False  # True: force the generic kernel everywhere (A/B timing)
# Split-K for launches whose tile grid cannot fill the chip (B = 1 / short utterances; conv_splitk.hip): every ConvOp
# is handed the per-device workspace below and the library decides per call.  SERENADE_AMD_SPLITK=0 turns it off.
SPLITK = os.environ.get("SERENADE_AMD_SPLITK", "1") != "0"
SPLITK_WS_BYTES = 32 << 20  # upper bound of the library's need (<= 192 tiles of 64 x 64, <= 8 slices, fp32)
_SPLITK_WS = {}
PROFILE = None  # set to a list by bench.py to collect (start event, end event, op) per conv_gemm launch


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(x):
    """tensor | (tensor, element_offset) | None -> raw device address (int) or None."""
    if x is None:
        return None
    if isinstance(x, tuple):
        t, off = x
        return t.data_ptr() + 4 * int(off)
    return x.data_ptr()


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32, "expected a CUDA fp32 tensor"
    return t


class ConvOp:
    """A prebuilt srn_conv_gemm call (parameters frozen, pointers borrowed from live tensors)."""

    __slots__ = ("p", "kw", "_fn", "_wplanes")

    def __init__(self, **kw):
        self.kw = kw  # kept for introspection (tests emulate the C-ABI contract from it) and to pin the tensors
        self._build(**kw)

    def _build(self, *, in0, w, out, n_batch, T_in, T_out, C_in, N, ld_in0, ldw, ld_out, taps=(0,), n_head=1,
                 in0_bs=0, in0_hs=0, in1=None, C_in0=0, in1_bs=0, ld_in1=0, C_w=0, w_bs=0, w_hs=0, w_nmajor=False,
                 bias=None, len_in=None, len_out=None, in_stride=1, reflect=False, pro_act=ACT_NONE, pro_slope=0.0,
                 alpha=1.0, beta=0.0, geglu=False, res=None, res_mode=RES_NONE, res_bs=0, res_hs=0, ld_res=0, res2=None,
                 res2_bs=0, ld_res2=0, post=POST_NONE, post_div=1.0, out_bs=0, out_hs=0, out_t_stride=1, out_t_off=0,
                 gn_partials=None, N_out=0, tile=0, precision=None, no_halo=False, out_tr=None, out_tr_bs=0,
                 ld_out_tr=0, out_tr_col0=0):
        p = SrnConvParams()
        p.n_batch, p.n_head, p.T_in, p.T_out = int(n_batch), int(n_head), int(T_in), int(T_out)
        p.C_in, p.C_in0, p.C_w, p.N, p.N_out = int(C_in), int(C_in0), int(C_w), int(N), int(N_out)
        taps = [int(t) for t in taps]
        assert 1 <= len(taps) <= _lib.SRN_MAX_TAPS
        p.n_taps = len(taps)
        for i, t in enumerate(taps):
            p.tap_off[i] = t
        p.in_stride, p.pad_reflect, p.w_nmajor = int(in_stride), int(reflect), int(bool(w_nmajor))  # reflect: 0 / 1 / 2
        p.pro_act, p.pro_slope, p.alpha, p.beta = int(pro_act), float(pro_slope), float(alpha), float(beta)
        p.geglu, p.res_mode, p.post, p.post_div = int(bool(geglu)), int(res_mode), int(post), float(post_div)
        p.out_t_stride, p.out_t_off, p.tile = int(out_t_stride), int(out_t_off), int(tile)
        p.in0, p.in0_bs, p.in0_hs, p.ld_in0 = _ptr(in0), int(in0_bs), int(in0_hs), int(ld_in0)
        p.in1, p.in1_bs, p.ld_in1 = _ptr(in1), int(in1_bs), int(ld_in1)
        p.w, p.w_bs, p.w_hs, p.ldw = _ptr(w), int(w_bs), int(w_hs), int(ldw)
        p.bias, p.len_in, p.len_out = _ptr(bias), _ptr(len_in), _ptr(len_out)
        p.res, p.res_bs, p.res_hs, p.ld_res = _ptr(res), int(res_bs), int(res_hs), int(ld_res)
        p.res2, p.res2_bs, p.ld_res2 = _ptr(res2), int(res2_bs), int(ld_res2)
        p.out, p.out_bs, p.out_hs, p.ld_out = _ptr(out), int(out_bs), int(out_hs), int(ld_out)
        p.gn_partials = _ptr(gn_partials)
        p.out_tr, p.out_tr_bs, p.ld_out_tr, p.out_tr_col0 = _ptr(out_tr), int(out_tr_bs), int(ld_out_tr), int(out_tr_col0)
        p.precision = int(DEFAULT_PRECISION if precision is None else precision)
        p.no_halo = 1 if NO_HALO else int(no_halo)  # 0 auto, 1 tiled kernels only, 2 force halo, 3 generic kernel only, 4 force strip
        self.p = p
        self._fn = _lib.lib().srn_conv_gemm
        self._wplanes = None
        if p.precision in (_lib.PREC_BF16X3, _lib.PREC_BF16X6):
            # static weights are split once, at plan-build time, into the bf16 plane images conv_fast.hip streams
            # straight into LDS: (hi | lo) for bf16x3; (hi | mid) + a separate lo plane for bf16x6
            if isinstance(w, torch.Tensor) and w.is_cuda and p.w_bs == 0 and p.w_hs == 0 and not w_nmajor:
                self._wplanes = weight_planes(w, p.N, p.n_taps, p.C_in, p.ldw, three=p.precision == _lib.PREC_BF16X6)
                p.w_hi = self._wplanes[0].data_ptr()
                if p.precision == _lib.PREC_BF16X6:
                    p.w_lo = self._wplanes[2].data_ptr()
        if SPLITK and isinstance(out, (torch.Tensor, tuple)):
            o = out[0] if isinstance(out, tuple) else out
            if o.is_cuda and self._fn and _lib.lib().srn_conv_gemm_workspace_bytes(ctypes.byref(p)) > 0:
                ws = splitk_workspace(o.device)  # one per device: the ops of a plan run back to back on one stream
                p.ws, p.ws_bytes = ws.data_ptr(), ws.numel()

    def __call__(self, stream=None):
        if PROFILE is not None and stream is None:
            # bench.py: HIP events on the launch stream around every conv_gemm launch
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            check(self._fn(ctypes.byref(self.p), _stream()), "srn_conv_gemm")
            e.record()
            PROFILE.append((s, e, self))
            return
        check(self._fn(ctypes.byref(self.p), stream if stream is not None else _stream()), "srn_conv_gemm")


def splitk_workspace(device):
    """the split-K partial-sum slab of the stream that is current while an op is BUILT: ops of one plan run back to
    back on the stream they were built under, and two streams never share a slab (plans driven from different streams
    would overwrite each other's partial sums).  Built under a graph capture, the slab comes from the graph's pool
    and is keyed by the capture stream, so eager code cannot pick it up."""
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (str(device), stream)
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=device)
    return _SPLITK_WS[key]


class ResUnitOp:
    """A prebuilt srn_hifigan_resunit call: one fused HiFi-GAN residual unit (see include/serenade_hip.h)."""

    __slots__ = ("p", "kw", "_fn", "_wplanes")

    def __init__(self, **kw):
        self.kw = kw
        self._build(**kw)

    def _build(self, *, x, w1, b1, w2, b2, out, n_batch, T, C, k, dilation, slope, res2=None, post_div=0.0,
               precision=None):
        p = SrnResUnitParams()
        p.n_batch, p.T, p.C, p.k, p.dilation, p.slope = int(n_batch), int(T), int(C), int(k), int(dilation), float(slope)
        p.x, p.x_bs = _ptr(x), int(T) * int(C)
        p.w1, p.b1, p.w2, p.b2 = _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2)
        p.res2, p.res2_bs = _ptr(res2), int(T) * int(C)
        p.post_div = float(post_div)
        p.out, p.out_bs = _ptr(out), int(T) * int(C)
        p.precision = int(DEFAULT_PRECISION if precision is None else precision)
        self._wplanes = None
        if p.precision == _lib.PREC_BF16X3 and w1.is_cuda:
            self._wplanes = (weight_planes(w1, C, k, C, k * C), weight_planes(w2, C, k, C, k * C))
            p.w1_hi, p.w2_hi = self._wplanes[0][0].data_ptr(), self._wplanes[1][0].data_ptr()
        self.p = p
        self._fn = _lib.lib().srn_hifigan_resunit

    def __call__(self, stream=None):
        if PROFILE is not None and stream is None:  # bench.py: timed with the other contraction launches
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            check(self._fn(ctypes.byref(self.p), _stream()), "srn_hifigan_resunit")
            e.record()
            PROFILE.append((s, e, self))
            return
        check(self._fn(ctypes.byref(self.p), stream if stream is not None else _stream()), "srn_hifigan_resunit")


class MultiCopyOp:
    """srn_multi_copy: many small device-to-device copies as one launch per 160 entries -- src tensor i (contiguous
    fp32) goes to dst.view(-1)[offs[i] : offs[i] + src.numel()].  The table travels in the kernel arguments, so the call
    can sit inside a captured hipGraph."""

    __slots__ = ("srcs", "offs", "dst", "_lists", "_fn")

    def __init__(self, srcs, offs, dst):
        self.srcs, self.offs, self.dst = list(srcs), [int(o) for o in offs], dst
        self._fn = _lib.lib().srn_multi_copy
        self._lists = []
        for i0 in range(0, len(self.srcs), _lib.SRN_COPY_LIST_MAX):
            lst = _lib.SrnCopyList()
            chunk = self.srcs[i0:i0 + _lib.SRN_COPY_LIST_MAX]
            lst.n = len(chunk)
            for j, t in enumerate(chunk):
                assert t.is_contiguous() and t.dtype == torch.float32
                lst.src[j], lst.off[j], lst.len[j] = t.data_ptr(), self.offs[i0 + j], t.numel()
            self._lists.append(lst)

    def __call__(self, stream=None):
        st = stream if stream is not None else _stream()
        for lst in self._lists:
            check(self._fn(ctypes.byref(lst), _ptr(self.dst), st), "srn_multi_copy")


class TransposeMultiOp:
    """srn_transpose_multi: a table of batched 2-D transposes as one launch per 40 entries.  entries: (src, dst, B, R,
    Cc, src_bs, ld_src, dst_bs, ld_dst) with dst[b][c][r] = src[b][r][c]; src / dst: tensor or (tensor, element offset)."""

    __slots__ = ("entries", "_lists", "_fn")

    def __init__(self, entries):
        self.entries = list(entries)
        self._fn = _lib.lib().srn_transpose_multi
        self._lists = []
        for i0 in range(0, len(self.entries), _lib.SRN_TR_LIST_MAX):
            lst = _lib.SrnTransposeList()
            chunk = self.entries[i0:i0 + _lib.SRN_TR_LIST_MAX]
            lst.n = len(chunk)
            for j, (src, dst, B, R, Cc, src_bs, ld_src, dst_bs, ld_dst) in enumerate(chunk):
                lst.src[j], lst.dst[j] = _ptr(src), _ptr(dst)
                lst.B[j], lst.R[j], lst.Cc[j] = B, R, Cc
                lst.src_bs[j], lst.ld_src[j], lst.dst_bs[j], lst.ld_dst[j] = src_bs, ld_src, dst_bs, ld_dst
            self._lists.append(lst)

    def __call__(self, stream=None):
        st = stream if stream is not None else _stream()
        for lst in self._lists:
            check(self._fn(ctypes.byref(lst), st), "srn_transpose_multi")


_TN_WS = {}


_TN_OUTGROWN = []


def tn_workspace(device, nbytes):
    """slices of a time-sliced srn_tn_gemm: per (device, stream) like the split-K slab, grown on demand.  An outgrown
    slab is kept alive, not freed: a captured graph (training.GraphedStep) may have its address baked in, and the ops
    that took it are transient.  Growth is by doubling at least, so the kept slabs sum to less than the live one."""
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (str(device), stream)
    old = _TN_WS.get(key)
    if old is None or old.numel() < nbytes:
        if old is not None:
            _TN_OUTGROWN.append(old)
        _TN_WS[key] = torch.empty(max(nbytes, 64 << 20, 2 * (old.numel() if old is not None else 0)),
                                  dtype=torch.uint8, device=device)
    return _TN_WS[key]


class TnGemmOp:
    """srn_tn_gemm (include/serenade_hip.h): out[z, m, j*N + n] = alpha * sum_{item,t} a[z,item,t,m] *
    b[z,item,t*stride + shifts[j], n].  a / b / out: tensor or (tensor, element offset).  len_b (int32, per item): rows
    of b at or past it read as zero; colsum (M,): alpha * the column sums of a (a conv's bias gradient), same launch."""

    __slots__ = ("p", "kw", "_fn", "_ws")

    def __init__(self, *, a, b, out, n_items, T_a, T_b, M, N, lda, ldb, ldc, shifts=(0,), stride=1, n_batch=1,
                 n_head=1, a_bs=0, a_hs=0, a_is=0, b_bs=0, b_hs=0, b_is=0, out_bs=0, out_hs=0, alpha=1.0, n_inner=1,
                 a_is2=0, b_is2=0, len_b=None, colsum=None):
        self.kw = dict(a=a, b=b, out=out, n_items=n_items, T_a=T_a, T_b=T_b, M=M, N=N, lda=lda, ldb=ldb, ldc=ldc,
                       shifts=tuple(int(v) for v in shifts), stride=stride, n_batch=n_batch, n_head=n_head, a_bs=a_bs,
                       a_hs=a_hs, a_is=a_is, b_bs=b_bs, b_hs=b_hs, b_is=b_is, out_bs=out_bs, out_hs=out_hs, alpha=alpha,
                       n_inner=n_inner, a_is2=a_is2, b_is2=b_is2, len_b=len_b, colsum=colsum)
        p = _lib.SrnTnGemmParams()
        p.n_batch, p.n_head, p.n_items, p.T_a, p.T_b = n_batch, n_head, n_items, T_a, T_b
        p.stride, p.n_shifts, p.M, p.N = stride, len(shifts), M, N
        for i, v in enumerate(shifts):
            p.shift[i] = int(v)
        p.a, p.a_bs, p.a_hs, p.a_is, p.lda = _ptr(a), a_bs, a_hs, a_is, lda
        p.b, p.b_bs, p.b_hs, p.b_is, p.ldb = _ptr(b), b_bs, b_hs, b_is, ldb
        p.out, p.out_bs, p.out_hs, p.ldc, p.alpha = _ptr(out), out_bs, out_hs, ldc, float(alpha)
        p.n_inner, p.a_is2, p.b_is2 = n_inner, a_is2, b_is2
        if len_b is not None and len_b.dtype != torch.int32:
            raise TypeError("TnGemmOp: len_b must be int32")
        p.len_b, p.colsum = _ptr(len_b), _ptr(colsum)
        self._fn = _lib.lib().srn_tn_gemm
        need = int(_lib.lib().srn_tn_gemm_workspace_bytes(ctypes.byref(p)))
        self._ws = None
        if need:
            o = out[0] if isinstance(out, tuple) else out
            self._ws = tn_workspace(o.device, need)
            p.ws, p.ws_bytes = self._ws.data_ptr(), self._ws.numel()
        self.p = p

    def __call__(self, stream=None):
        check(self._fn(ctypes.byref(self.p), stream if stream is not None else _stream()), "srn_tn_gemm")


_WPLANES = {}  # (data_ptr, version, shape, ...) -> (planes, weight) bf16 weight planes, split once per weight VALUE


def clear_caches():
    """Drop the weight-plane cache (long-running services that load many checkpoints).
    Ops already built keep their own references, so existing plans stay valid."""
    _WPLANES.clear()


def drop_weight_planes(tensors):
    """Forget the cached planes of the given weight tensors (a module re-packing its weights calls this, so planes
    of a previous checkpoint can never be contracted with the new one even when the storage address is reused)."""
    ptrs = {t.data_ptr() for t in tensors if isinstance(t, torch.Tensor)}
    for key in [k for k in _WPLANES if k[0] in ptrs]:
        del _WPLANES[key]


def weight_planes(w, N, n_taps, C_in, ldw, three=False):
    """fp32 packed weights [N][n_taps * C_in] -> bf16 planes [N][n_taps][roundup(C_in, 32) / 32][hi 32 | lo 32],
    hi = bf16(w), lo = bf16(w - hi) (round-to-nearest-even, the same split the kernels apply to activations).
    three=True (bf16x6): the exact three-way split w = hi + mid + lo; returns (planes [..][hi 32 | mid 32], w,
    lo plane [N][n_taps][chunks][32]).
    The cache key carries the tensor's in-place version counter: `load_state_dict` copies into the live parameter
    (same address, version + 1), which therefore misses the cache and is split again."""
    try:
        version = w._version
    except RuntimeError:  # tensors created under torch.inference_mode() carry no counter (and cannot be written)
        version = -1
    key = (w.data_ptr(), version, tuple(w.shape), N, n_taps, C_in, ldw, bool(three))
    if key not in _WPLANES:
        for stale in [k for k in _WPLANES if k[0] == key[0] and k[2:] == key[2:]]:
            del _WPLANES[stale]  # an older value of the same tensor
        cp = (C_in + 31) // 32 * 32
        src = w.reshape(-1)[: N * ldw].view(N, ldw)[:, : n_taps * C_in].reshape(N, n_taps, C_in)
        full = torch.zeros(N, n_taps, cp, device=w.device, dtype=torch.float32)
        full[:, :, :C_in] = src
        hi = full.to(torch.bfloat16)
        r1 = full - hi.to(torch.float32)
        lo = r1.to(torch.bfloat16)  # bf16x3: second plane; bf16x6: the MID plane
        # kernel layout: per row, per 32-channel chunk, [32 hi | 32 lo]  (one 128-B line per chunk)
        pl = torch.stack([hi.view(N, n_taps, cp // 32, 32), lo.view(N, n_taps, cp // 32, 32)], dim=3).contiguous()
        if three:
            third = (r1 - lo.to(torch.float32)).to(torch.bfloat16).view(N, n_taps, cp // 32, 32).contiguous()
            _WPLANES[key] = (pl, w, third)
        else:
            _WPLANES[key] = (pl, w)  # keep `w` alive so the data_ptr key stays unique
    return _WPLANES[key]


class CallOp:
    """A prebuilt call of any other C-ABI function (stream appended at run time)."""

    __slots__ = ("_fn", "_args", "targs", "name")

    def __init__(self, name, targs):
        self._fn = getattr(_lib.lib(), name)
        self.targs = tuple(targs)  # python-level arguments (tensors / scalars), pins the tensors
        self._args = tuple(_ptr(a) if (a is None or isinstance(a, (torch.Tensor, tuple))) else a for a in targs)
        self.name = name

    def __call__(self, stream=None):
        check(self._fn(*self._args, stream if stream is not None else _stream()), self.name)


# ------------------------------------------------------------------ op builders (return CallOp)
def gn_mish_apply_op(x, partials, gamma, beta, time_bias, lens, y, B, T, C, groups=8, eps=1e-5, tb_bs=0,
                     valid_stats=False):
    return CallOp("srn_gn_mish_apply", (x, partials, gamma, beta, time_bias, tb_bs, lens, y, B, T, C, groups, eps,
                                        int(bool(valid_stats))))


def resblock_tail_op(c2, partials, gamma, beta, lens, r, scale, shift, ld_ss, y, B, T, C, groups=8, gn_eps=1e-5,
                     ln_eps=1e-5, valid_stats=False):
    return CallOp("srn_resblock_tail", (c2, partials, gamma, beta, lens, r, scale, shift, ld_ss, y, B, T, C, groups,
                                        gn_eps, ln_eps, int(bool(valid_stats))))


def resblock_tail_ln_op(c2, partials, gamma, beta, lens, r, scale, shift, ld_ss, y, ln_w, ln_b, y2, B, T, C, groups=8,
                        gn_eps=1e-5, ln_eps=1e-5, valid_stats=False, ln2_eps=1e-5):
    """resblock_tail_op + the LayerNorm that opens the transformer block behind it, in one launch"""
    return CallOp("srn_resblock_tail_ln", (c2, partials, gamma, beta, lens, r, scale, shift, ld_ss, y, B, T, C, groups,
                                           gn_eps, ln_eps, int(bool(valid_stats)), ln_w, ln_b, y2, ln2_eps))


def scatter_rows_op(src, src_bs, ld_src, dst, dst_bs, ld_dst, dc0, row_off, n_rows, B, T, C):
    return CallOp("srn_scatter_rows", (src, src_bs, ld_src, dst, dst_bs, ld_dst, dc0, row_off, n_rows, B, T, C))


def layernorm_op(x, gamma, beta, y, rows, C, eps=1e-5):
    return CallOp("srn_layernorm", (x, gamma, beta, y, rows, C, eps))


def softmax_rows_op(s, lens, Z, n_head, L, ld):
    return CallOp("srn_softmax_rows", (s, lens, Z, n_head, L, ld))


def sinusoidal_emb_op(t, out, n, dim, ld, scale=1000.0):
    return CallOp("srn_sinusoidal_emb", (t, out, n, dim, ld, scale))


def copy_channels_op(src, src_bs, ld_src, sc0, dst, dst_bs, ld_dst, dc0, B, T, C):
    return CallOp("srn_copy_channels", (src, src_bs, ld_src, sc0, dst, dst_bs, ld_dst, dc0, B, T, C))


def transpose_op(src, dst, B, R, Cc, src_bs, ld_src, dst_bs, ld_dst):
    """dst[b][c][r] = src[b][r][c]"""
    return CallOp("srn_transpose_ct", (src, dst, B, R, Cc, src_bs, ld_src, dst_bs, ld_dst))


def renorm_op(x, trg_scale, trg_mean, voc_mean, voc_scale, y, rows, C):
    return CallOp("srn_renorm", (x, trg_scale, trg_mean, voc_mean, voc_scale, y, rows, C))


def out_conv_tanh_op(x, w, bias, y, B, T, C, k, slope):
    return CallOp("srn_out_conv_tanh", (x, w, bias, y, B, T, C, k, slope))


def pd_gather_op(x, d, out, B, T, C, dilation, slope):
    return CallOp("srn_pd_gather", (x, d, out, B, T, C, float(dilation), float(slope)))


def gru_recur_last_op(gi, w_hh_t, b_hh, h, B, T, H):
    return CallOp("srn_gru_recur_last", (gi, w_hh_t, b_hh, h, B, T, H))


def style_token_attention_kv_op(ref, wq_t, bq, k, v, wo_t, bo, out, B, Dq, n_tok, F, n_head):
    return CallOp("srn_style_token_attention_kv", (ref, wq_t, bq, k, v, wo_t, bo, out, B, Dq, n_tok, F, n_head))


# ------------------------------------------------------------------ weight packing (one-time, at load)
def pack_conv_weight(w, c_pad=None):
    """torch Conv1d weight (Co, Ci, k) -> [Co][k][Ci_pad] k-contiguous GEMM operand."""
    co, ci, k = w.shape
    cp = ci if c_pad is None else c_pad
    out = w.new_zeros(co, k, cp)
    out[:, :, :ci] = w.permute(0, 2, 1)
    return out.reshape(co, k * cp).contiguous()


def conv_taps(k, dilation=1, padding=None):
    """input-row offsets of a stride-1 Conv1d: tap j reads t + j*dilation - padding."""
    if padding is None:
        padding = (k - 1) // 2 * dilation
    return [j * dilation - padding for j in range(k)]


def convtranspose_phases(w, stride, padding):
    """torch ConvTranspose1d weight (Ci, Co, k) -> per output phase r (out index = m*stride + r):
    (taps = input-row offsets relative to m, packed weight [Co][n_taps*Ci]).
    out[o] = sum_i sum_k x[i] w[:, :, k] with o = i*stride - padding + k."""
    ci, co, k = w.shape
    phases = []
    for r in range(stride):
        taps, mats = [], []
        # k = r + padding + j*stride for integer j, 0 <= k < kernel;  i = m - j
        j = -((r + padding) // stride)
        while True:
            kk = r + padding + j * stride
            if kk >= k:
                break
            if kk >= 0:
                taps.append(-j)
                mats.append(w[:, :, kk].t())  # (Co, Ci)
            j += 1
        packed = torch.stack(mats, dim=1).reshape(co, len(taps) * ci).contiguous()
        phases.append((taps, packed))
    return phases


def pack_geglu(w, b):
    """GEGLU proj (2*inner, dim): rows [0, inner) = value, [inner, 2 inner) = gate  ->  rows interleaved in
    32-row (value | gate) groups so one 64-wide MFMA column span holds both halves of 32 outputs."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0
    idx = torch.arange(inner, device=w.device).view(-1, 32)
    order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    return w[order].contiguous(), b[order].contiguous()


# ------------------------------------------------------------------ hipGraph replay of a fixed op list
GRAPHS = os.environ.get("SERENADE_AMD_GRAPHS", "0") == "1"


def set_graphs(enabled):
    """Replay plans (the per-utterance-batch op lists) as captured hipGraphs.  Pays off when launches, not kernels,
    bound the step: B=1 / short utterances (~1600 launches of 5-20 us each)."""
    global GRAPHS
    GRAPHS = bool(enabled)


class GraphRunner:
    """Runs `get_ops()` (a flat list of ConvOp / CallOp over static buffers) eagerly, or -- with GRAPHS on -- as one
    captured hipGraph.  The first call is always eager (every kernel sets its attributes and loads its code object
    outside capture); per-launch profiling (PROFILE) always runs eagerly.  `invalidate()` after any change of the op
    list or of a buffer address baked into it."""

    def __init__(self, get_ops):
        self._get_ops = get_ops
        self._graph = None
        self._warm = False

    def invalidate(self):
        self._graph = None
        self._warm = False

    def __call__(self):
        op_list = self._get_ops()
        if not GRAPHS or PROFILE is not None or not self._warm or not torch.cuda.is_available():
            for op in op_list:
                op()
            self._warm = True
            return
        if self._graph is None:
            g = torch.cuda.CUDAGraph()
            # callers run under torch.inference_mode() (flow_matching.py:39); capture bookkeeping (generator state
            # registration) must happen outside it.  Only C-ABI launches are recorded, no torch ops.
            with torch.inference_mode(False), torch.cuda.graph(g):
                for op in op_list:
                    op()
            self._graph = g
        self._graph.replay()
