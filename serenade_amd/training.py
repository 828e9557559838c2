"""Training step on MI355X.

What the reference does per step (trainers/ssc.py:57-96, bin/ssc_train.py:331-359): `Serenade.forward` -> cfm_loss +
prior_loss -> `backward()` -> DDP gradient all-reduce over NCCL -> `clip_grad_norm_(1.0)` -> AdamW(lr 8e-4).  Here:

  * `TrainSerenade`  the whole model (serenade.py:35-166) for training: every trainable tensor of the checkpoint under
                     its reference name, all views of ONE flat fp32 buffer (and one flat gradient buffer);
  * `Estimator`      the flow-matching decoder (matcha_components/decoder.py, 192 tensors, >99 % of the FLOPs): forward
                     over the same channels-last HIP kernels as inference, every op an autograd node whose backward is
                     HIP again; `cfm_loss` = flow_matching.py:95-133 around it;
  * `GradSync`       the DDP replacement: the flat gradient buffer is all-reduced over RCCL in buckets that are
                     launched from backward hooks as soon as their parameters are done (overlaps the rest of backward);
  * `AdamW`          clip_grad_norm_ + torch.optim.AdamW as one fused kernel launch over the flat buffers (srn_adamw).

Division of labour.  GEMM-shaped gradients that contract over channels -- dgrad of every conv / projection (a conv of
dY with the tap-reversed, transposed weights), dP = dO V^T and dQ = dS K of attention -- are `srn_conv_gemm` launches;
row / column reductions and activations (GroupNorm+Mish, LayerNorm / SpeakerAdapter, softmax, GEGLU) are the kernels of
csrc/train.hip.  Gradients that contract over TIME (wgrad = dY^T X, dV = P^T dO, dK = dS^T Q) are plain transposed-A
GEMMs and go to rocBLAS through `torch.matmul` (the library-GEMM case).  torch autograd is the tape; mask multiplies,
concatenations, LeakyReLU / dropout, weight-norm folding, the (B, 2048) time-embedding activations and the scalar loss
reductions are torch ops on the same device.  The GST style encoder (0.8 GFLOP of a ~1.3 TFLOP step: 3x3 Conv2d +
train-mode BatchNorm2d, a 16-step GRU, 50-token attention) runs on torch's GPU ops in both directions.  Dropout uses
torch's generator (the reference's draws cannot be reproduced); parity tests run with dropout off against the
reference's own gradients (tests/golden/train_grads_L45.npz, train_full_T64.npz).
"""
import math

import torch
import torch.nn.functional as F

from . import _lib, ops
from .ops import TnGemmOp, ConvOp

__all__ = ["Estimator", "TrainSerenade", "ParamStore", "GraphedStep", "MultiStepLR", "save_checkpoint", "load_checkpoint", "GradSync", "AdamW", "cfm_loss", "conv1d", "gn_mish", "row_ln", "attention_core", "geglu"]


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} needs the HIP library and an MI355X tensor (there is no CPU fallback)")


def _call(name, *args):
    ops.CallOp(name, args)()


NORM_BWD_ROWS = 8  # rows per chunk of partial sums in the LayerNorm / GroupNorm backward kernels (srn_rowln_chunks, srn_gn_chunks)


def _rup(n, m):
    return (n + m - 1) // m * m


# =====================================================================================================================
#  conv / linear
# =====================================================================================================================
def _launch_conv(x, w, bias, y, taps, B, T_in, T_out, C, N, in_stride=1, out_t_stride=1, out_t_off=0, ld_out=None,
                 out_bs=None, gn_partials=None, len_in=None, len_out=None):
    ConvOp(in0=x, w=w, out=y, n_batch=B, T_in=T_in, T_out=T_out, C_in=C, N=N, in0_bs=T_in * C, ld_in0=C,
           ldw=w.shape[1], out_bs=(T_out * N if out_bs is None else out_bs), ld_out=(N if ld_out is None else ld_out),
           bias=bias, taps=taps, in_stride=in_stride, out_t_stride=out_t_stride, out_t_off=out_t_off,
           gn_partials=gn_partials, len_in=len_in, len_out=len_out, precision=_lib.PREC_FP32)()


class _Conv(torch.autograd.Function):
    """y[b, t, :] = sum_j x[b, t * stride + taps[j], :] W_j^T + bias  (rows outside [0, T) read as zero).
    x (B, T, C) fp32 contiguous, C % 4 == 0; w packed (N, len(taps) * C), k-major; optional GroupNorm partial sums of
    the output (32 x 32 tiles, conv epilogue) as a second, non-differentiable result.
    lens (B,) int32 or None (stride 1 only): input rows t >= lens[b] count as zero -- the reference's `x * mask` in front
    of its convs (decoder.py:66-101) without the multiply: the forward reads them as zero (len_in), dX's rows past the
    length are stored as zero (len_out), and the weight gradient skips them (srn_tn_gemm's len_b)."""

    @staticmethod
    def forward(ctx, x, w, bias, taps, stride, T_out, want_gn, lens=None, wd=None):
        _require_cuda(x, "training.conv1d")
        B, T, C = x.shape
        N = w.shape[0]
        x, w = x.contiguous(), w.contiguous()
        y = torch.empty(B, T_out, N, device=x.device, dtype=torch.float32)
        part = None
        if want_gn:
            part = torch.empty(B, (T_out + 31) // 32, N // 32, 2, device=x.device, dtype=torch.float32)
        assert lens is None or stride == 1
        ctx.set_materialize_grads(False)  # no zero-filled gradient tensor for the non-differentiable partial sums
        _launch_conv(x, w, bias, y, taps, B, T, T_out, C, N, in_stride=stride, gn_partials=part, len_in=lens)
        ctx.save_for_backward(x, w, wd)  # wd too: a re-laid buffer of the estimator, version-checked like w
        ctx.taps, ctx.stride, ctx.has_bias, ctx.lens = tuple(taps), stride, bias is not None, lens
        if want_gn:
            ctx.mark_non_differentiable(part)
            return y, part
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        if dy is None:
            return (None,) * 9
        x, w, wd_saved = ctx.saved_tensors
        taps, stride, lens = ctx.taps, ctx.stride, ctx.lens
        B, T, C = x.shape
        _, T_out, N = dy.shape
        nt = len(taps)
        dy = dy.contiguous()
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0]:
            # dgrad: dX[t] = sum_j dY[(t - taps[j]) / stride] W_j over the taps that divide -> a conv of dY with the
            # transposed weights Wd[c][j][n] = W[n][j][c]; with stride 2 one launch per output-row parity
            if wd_saved is not None:  # re-laid once for the whole step (Estimator._relay)
                wd = wd_saved.view(C, nt, N)
            else:
                wd = torch.empty(C, nt, N, device=dy.device, dtype=torch.float32)  # LDS tile transposes, one batch per tap
                _call("srn_transpose_ct", w, wd, nt, N, C, C, nt * C, N, nt * N)
            dx = torch.empty(B, T, C, device=dy.device, dtype=torch.float32)
            if stride == 1:
                _launch_conv(dy, wd.view(C, nt * N), None, dx, [-o for o in taps], B, T_out, T, N, C, len_out=lens)
            else:
                for ph in range(stride):
                    sel = [j for j, o in enumerate(taps) if (ph - o) % stride == 0]
                    rows = (T - ph + stride - 1) // stride
                    if rows <= 0:
                        continue
                    if not sel:
                        dx[:, ph::stride] = 0
                        continue
                    wsel = torch.stack([wd[:, j, :] for j in sel], dim=1).reshape(C, len(sel) * N)  # no host-side index tensor
                    _launch_conv(dy, wsel, None, dx, [(ph - taps[j]) // stride for j in sel], B, T_out, rows, N, C,
                                 out_t_stride=stride, out_t_off=ph, ld_out=C, out_bs=T * C)
        if ctx.needs_input_grad[1]:
            # wgrad: dW_j = sum_{b,t} dY[b,t,:]^T x[b, t*stride + taps[j], :] -- a contraction over time on operands
            # that are time-major as they lie: srn_tn_gemm (rows outside the item read as zero; time sliced over
            # workgroups, slices added in order), straight into the packed (N, taps * C) layout; the bias gradient (the
            # column sums of dY) rides along in the same launch
            dw = torch.empty(N, nt * C, device=dy.device, dtype=torch.float32)
            if want_db and N % 4 == 0:
                db = torch.empty(N, device=dy.device, dtype=torch.float32)
            TnGemmOp(a=dy, b=x, out=dw, n_items=B, T_a=T_out, T_b=T, M=N, N=C, lda=N, ldb=C, ldc=nt * C, shifts=taps,
                     stride=stride, a_is=T_out * N, b_is=T * C, len_b=lens, colsum=db)()
        if want_db and db is None:
            db = _colsum(dy.reshape(-1, N))  # own kernel, not a torch reduction: see AdamW._grad_norm
        return dx, dw, db, None, None, None, None, None, None


def conv1d(x, w, bias, taps=(0,), stride=1, want_gn=False, T_out=None, lens=None, wd=None):
    """channels-last conv / linear over the HIP contraction kernel.  x (B, T, C) or (rows, C); w packed (N, k * C).
    T_out (stride 1 only): number of output rows when the input was padded by the caller (taps >= 0, "valid" conv).
    lens (B,) int32 (stride 1 only): conv of x with its rows past lens[b] zeroed (`x * mask`), see _Conv.
    wd: the weights already laid out for the input gradient, (C, taps * N) with wd[c, j * N + n] = w[n, j * C + c], when
    the caller keeps them (Estimator._relay); made on the fly otherwise."""
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    T = x.shape[1]
    if T_out is not None:
        assert stride == 1 and max(taps) + T_out <= T
    elif stride == 1:
        T_out = T
    else:  # torch Conv1d with padding (k - 1) / 2: taps -p .. p
        p = -min(taps)
        T_out = (T + 2 * p - (len(taps) - 1) - 1) // stride + 1
    out = _Conv.apply(x, w, bias, tuple(int(t) for t in taps), int(stride), int(T_out), bool(want_gn), lens, wd)
    if two_d:
        return out.squeeze(0)
    return out


def _colsum(x):
    """column sums over the rows of a contiguous (R, N) matrix -> (N,), or per item of (B, R, N) -> (B, N)
    (srn_colsum: two small launches, fixed summation order)"""
    two_d = x.dim() == 2
    B, R, N = (1, *x.shape) if two_d else x.shape
    part = torch.empty(B, (R + 31) // 32, N, device=x.device, dtype=torch.float32)  # srn_colsum_chunks(R)
    out = torch.empty(B, N, device=x.device, dtype=torch.float32)
    _call("srn_colsum", x, part, out, B, R, N, N)
    return out[0] if two_d else out


class _AddRowBias(torch.autograd.Function):
    """h (B, T, C) + b (B, C) broadcast over T.  Its backward's column sums are srn_colsum launches instead of the
    reduction autograd would insert for the broadcast: see AdamW._grad_norm."""

    @staticmethod
    def forward(ctx, h, b):
        return h + b.unsqueeze(1)

    @staticmethod
    def backward(ctx, dy):
        B, T, C = dy.shape
        dy = dy.contiguous()
        return dy, _colsum(dy)


class _PackConv(torch.autograd.Function):
    """(N, C, k) -> (N, k * C_pad): per output channel a (C, k) -> (k, C) LDS tile transpose (srn_transpose_ct), zero pad
    columns; backward the transpose back.  (torch's strided copy of the same permute ran at < 1 TB/s and there are
    three of them per weight per step: pack, its gradient, and the dgrad operand.)"""

    @staticmethod
    def forward(ctx, w, c_pad):
        n, c, k = w.shape
        ctx.dims = (n, c, k, c_pad)
        w = w.contiguous()
        out = (torch.zeros if c_pad > c else torch.empty)(n, k * c_pad, device=w.device, dtype=torch.float32)
        _call("srn_transpose_ct", w, out, n, c, k, c * k, k, k * c_pad, c_pad)
        return out

    @staticmethod
    def backward(ctx, dp):
        n, c, k, c_pad = ctx.dims
        dw = torch.empty(n, c, k, device=dp.device, dtype=torch.float32)
        _call("srn_transpose_ct", dp.contiguous(), dw, n, k, c, k * c_pad, c_pad, c * k, k)
        return dw, None


class _PackAll(torch.autograd.Function):
    """Every conv weight of the estimator (N, C, k) -> (N, k * C_pad) in ONE launch, into buffers the estimator keeps
    (Estimator._relay), together with W^T of every conv / linear for the input gradients; backward: the packed gradients
    back to (N, C, k), one launch again.  Issued one by one these were ~140 launches of a few microseconds per step."""

    @staticmethod
    def forward(ctx, est, *weights):
        for op in est._relay_ops:
            op()
        # the launches above rewrote the persistent buffers through raw pointers: tell autograd, so that a backward of an
        # EARLIER forward (which saved aliases of them) raises instead of mixing its activations with these weights
        for t in (*est._pk.values(), *est._wd.values()):
            torch.autograd.graph.increment_version(t)
        ctx.est = est
        return tuple(est._pk[name].detach() for name in est._pack_names)  # fresh aliases: apply() marks its outputs

    @staticmethod
    def backward(ctx, *dps):
        est = ctx.est
        outs, entries = [], []
        for name, dp in zip(est._pack_names, dps):
            n, c, k = est.params[name].shape
            if dp is None:
                outs.append(None)
                continue
            c_pad = est._pk[name].shape[1] // k
            dp = dp.contiguous()
            dw = torch.empty(n, c, k, device=dp.device, dtype=torch.float32)
            entries.append((dp, dw, n, k, c, k * c_pad, c_pad, c * k, k))
            outs.append(dw)
        if entries:
            ops.TransposeMultiOp(entries)()
        return (None, *outs)


class _WeightNormPack(torch.autograd.Function):
    """torch weight_norm (w = g v / ||v|| per output channel) of a conv weight (N, C, k) straight into the packed k-major
    rows (N, k * C) the conv kernels read, plus W^T for the input gradient (`.wd` of the ctx-less return: second output,
    non-differentiable): srn_weight_norm_fwd / _bwd, one launch per direction instead of ~14 torch launches and three
    transposes per conv."""

    @staticmethod
    def forward(ctx, v, g):
        n, c, k = v.shape
        v, g = v.contiguous(), g.contiguous()
        w = torch.empty(n, k * c, device=v.device, dtype=torch.float32)
        wd = torch.empty(c, k * n, device=v.device, dtype=torch.float32)
        inv = torch.empty(n, device=v.device, dtype=torch.float32)
        _call("srn_weight_norm_fwd", v, g, w, wd, inv, n, c, k)
        ctx.save_for_backward(v, g, inv)
        ctx.mark_non_differentiable(wd)
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" of wd
        return w, wd

    @staticmethod
    def backward(ctx, dw, _):
        if dw is None:
            return None, None
        v, g, inv = ctx.saved_tensors
        n, c, k = v.shape
        dv = torch.empty_like(v)
        dg = torch.empty(n, device=v.device, dtype=torch.float32)
        _call("srn_weight_norm_bwd", dw.contiguous(), v, g, inv, dv, dg, n, c, k)
        return dv, dg.view(g.shape)


def pack_conv(w, c_pad=None):
    """torch Conv1d weight (N, C, k) -> (N, k * C_pad), differentiable."""
    n, c, k = w.shape
    c_pad = c if c_pad is None else max(int(c_pad), c)
    if k == 1 and c_pad == c:
        return w.reshape(n, c)  # a view
    return _PackConv.apply(w, c_pad)


# =====================================================================================================================
#  GroupNorm -> Mish -> mask   (Block1D, decoder.py:66-77)
# =====================================================================================================================
def _gn_stats(part, T, C, groups, eps):
    """(mean, rstd) (B, groups) fp32 from the conv epilogue's 32 x 32 tile sums; statistics over the padded length"""
    B = part.shape[0]
    mean = torch.empty(B, groups, device=part.device, dtype=torch.float32)
    rstd = torch.empty(B, groups, device=part.device, dtype=torch.float32)
    _call("srn_gn_stats", part, mean, rstd, B, T, C, groups, eps)
    return mean, rstd


class _GNMish(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, part, gamma, beta, lens, groups, eps):
        B, T, C = h.shape
        y = torch.empty_like(h)
        _call("srn_gn_mish_apply", h, part, gamma, beta, None, 0, lens, y, B, T, C, groups, eps, 0)
        mean, rstd = _gn_stats(part, T, C, groups, eps)
        ctx.save_for_backward(h, mean, rstd, gamma, beta, lens)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        h, mean, rstd, gamma, beta, lens = ctx.saved_tensors
        G = ctx.groups
        B, T, C = h.shape
        dy = dy.contiguous()
        nch = (T + NORM_BWD_ROWS - 1) // NORM_BWD_ROWS
        part = torch.empty(B, nch, 2, C, device=h.device, dtype=torch.float32)
        _call("srn_gn_mish_bwd_partial", h, dy, mean, rstd, gamma, beta, lens, part, B, T, C, G)
        col = torch.empty(B, 2, C, device=h.device, dtype=torch.float32)  # sum_t dg, sum_t dg * xhat
        gsum = torch.empty(B, G, 2, device=h.device, dtype=torch.float32)
        _call("srn_chunk_colsum", part, gamma, col, gsum, B, nch, C, G)
        dcol = col.sum(0)
        dbeta, dgamma = dcol[0], dcol[1]
        dh = torch.empty_like(h)
        _call("srn_gn_mish_bwd_apply", h, dy, mean, rstd, gamma, beta, gsum, lens, dh, B, T, C, G)
        return dh, None, dgamma, dbeta, None, None, None


def gn_mish(h, part, gamma, beta, lens, groups=8, eps=1e-5):
    """mish(GroupNorm(h)) on rows < lens[b], zero after; `part` = the producing conv's GroupNorm partial sums."""
    return _GNMish.apply(h.contiguous(), part, gamma.contiguous(), beta.contiguous(), lens, groups, eps)


# =====================================================================================================================
#  per-frame LayerNorm with a per-(batch, channel) multiplier / offset: nn.LayerNorm and SpeakerAdapter
# =====================================================================================================================
class _RowLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, a, eps):
        B, T, C = x.shape
        per_b = m.dim() == 2
        y = torch.empty_like(x)
        _call("srn_rowln_fwd", x, m, C if per_b else 0, a, C if per_b else 0, y, B, T, C, eps)
        ctx.save_for_backward(x, m)
        ctx.eps, ctx.per_b = eps, per_b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, m = ctx.saved_tensors
        B, T, C = x.shape
        dy = dy.contiguous()
        nch = (T + NORM_BWD_ROWS - 1) // NORM_BWD_ROWS
        part = torch.empty(B, nch, 2, C, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        _call("srn_rowln_bwd", x, dy, m, C if ctx.per_b else 0, dx, part, B, T, C, ctx.eps)
        if ctx.per_b:
            col = torch.empty(B, 2, C, device=x.device, dtype=torch.float32)
            _call("srn_chunk_colsum", part, None, col, None, B, nch, C, 1)
            return dx, col[:, 0].contiguous(), col[:, 1].contiguous(), None
        col = torch.empty(2, C, device=x.device, dtype=torch.float32)  # shared weights: one sum over all items' chunks
        _call("srn_chunk_colsum", part, None, col, None, 1, B * nch, C, 1)
        return dx, col[0], col[1], None


def row_ln(x, m, a, eps=1e-5):
    """y = LayerNorm_C(x) * m + a; m, a (C,) (nn.LayerNorm) or (B, C) (SpeakerAdapter, decoder.py:34-45)."""
    return _RowLN.apply(x.contiguous(), m.contiguous(), a.contiguous(), eps)


# =====================================================================================================================
#  attention core: softmax(Q K^T / sqrt(d) + key mask) V on the fused (B, L, 3 * H * d) projection
# =====================================================================================================================
class _AttnCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, lens, H):
        B, L, three = qkv.shape
        inner = three // 3
        hd = inner // H
        Lp = _rup(L, 32)
        dev = qkv.device
        alpha = 1.0 / math.sqrt(hd)
        alloc = torch.empty if Lp == L else torch.zeros  # pad columns must read as zero
        P = alloc(B, H, L, Lp, device=dev, dtype=torch.float32)
        ConvOp(in0=qkv, w=(qkv, inner), out=P, n_batch=B, n_head=H, T_in=L, T_out=L, C_in=hd, N=L,
               in0_bs=L * three, in0_hs=hd, ld_in0=three, w_bs=L * three, w_hs=hd, ldw=three, out_bs=H * L * Lp,
               out_hs=L * Lp, ld_out=Lp, alpha=alpha, precision=_lib.PREC_FP32)()
        _call("srn_softmax_rows", P, lens, B * H, H, L, Lp)
        vt = alloc(B, inner, Lp, device=dev, dtype=torch.float32)  # V^T per head: (hd, Lp) k-major
        _call("srn_transpose_ct", (qkv, 2 * inner), vt, B, L, inner, L * three, three, inner * Lp, Lp)
        o = torch.empty(B, L, inner, device=dev, dtype=torch.float32)
        ConvOp(in0=P, w=vt, out=o, n_batch=B, n_head=H, T_in=L, T_out=L, C_in=Lp, N=hd, in0_bs=H * L * Lp,
               in0_hs=L * Lp, ld_in0=Lp, w_bs=inner * Lp, w_hs=hd * Lp, ldw=Lp, out_bs=L * inner, out_hs=hd,
               ld_out=inner, precision=_lib.PREC_FP32)()
        ctx.save_for_backward(qkv, P)
        ctx.H = H
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, P = ctx.saved_tensors
        H = ctx.H
        B, L, three = qkv.shape
        inner = three // 3
        hd = inner // H
        Lp = P.shape[-1]
        dev = qkv.device
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        # dP = dO V^T (rows of dO against rows of V: the Q K^T launch with other operands), then dS in place
        alloc = torch.empty if Lp == L else torch.zeros
        dS = alloc(B, H, L, Lp, device=dev, dtype=torch.float32)
        ConvOp(in0=do, w=(qkv, 2 * inner), out=dS, n_batch=B, n_head=H, T_in=L, T_out=L, C_in=hd, N=L,
               in0_bs=L * inner, in0_hs=hd, ld_in0=inner, w_bs=L * three, w_hs=hd, ldw=three, out_bs=H * L * Lp,
               out_hs=L * Lp, ld_out=Lp, precision=_lib.PREC_FP32)()
        _call("srn_softmax_bwd", P, dS, B * H * L, L, Lp, 1.0 / math.sqrt(hd))
        # dQ = dS K: contraction over keys, K^T per head as the k-major operand
        kt = alloc(B, inner, Lp, device=dev, dtype=torch.float32)
        _call("srn_transpose_ct", (qkv, inner), kt, B, L, inner, L * three, three, inner * Lp, Lp)
        ConvOp(in0=dS, w=kt, out=dqkv, n_batch=B, n_head=H, T_in=L, T_out=L, C_in=Lp, N=hd, in0_bs=H * L * Lp,
               in0_hs=L * Lp, ld_in0=Lp, w_bs=inner * Lp, w_hs=hd * Lp, ldw=Lp, out_bs=L * three, out_hs=hd,
               ld_out=three, precision=_lib.PREC_FP32)()
        # dK = dS^T Q, dV = P^T dO: contractions over query rows, one problem per (batch, head): srn_tn_gemm reads dS / P
        # (keys contiguous) and Q / dO (channels contiguous) as they lie and writes the head's slice of dqkv
        for src, rhs, col in ((dS, (qkv, 0), inner), (P, do, 2 * inner)):
            TnGemmOp(a=src, b=rhs, out=(dqkv, col), n_items=1, T_a=L, T_b=L, M=L, N=hd, lda=Lp,
                     ldb=(three if isinstance(rhs, tuple) else inner), ldc=three, n_batch=B, n_head=H,
                     a_bs=H * L * Lp, a_hs=L * Lp, b_bs=L * (three if isinstance(rhs, tuple) else inner), b_hs=hd,
                     out_bs=L * three, out_hs=hd)()
        return dqkv, None, None


def attention_core(qkv, lens, n_head):
    return _AttnCore.apply(qkv.contiguous(), lens, n_head)


# =====================================================================================================================
#  GST pieces: Conv2d(k3, s2, p1), BatchNorm2d(training) + ReLU, GRU (last state), style-token attention
# =====================================================================================================================
class _Conv2dS2(torch.autograd.Function):
    """y (B, Ho, Wo, Co) = Conv2d(k 3, stride 2, pad 1, no bias) of channels-last x (B, H, W, Cp) with the reference's
    weight (Co, Ci, 3, 3), Ci <= Cp.  srn_im2col_s2 gathers the receptive fields once ((B Ho Wo) x 9 Cp rows, padded to
    a multiple of 32 columns), then forward, dX (+ srn_col2im_s2) and dW (srn_tn_gemm) are ONE contraction each over
    all output positions -- the per-(b, ho) form of the inference path wastes its 64-row tiles once Wo <= 20."""

    @staticmethod
    def forward(ctx, x, w):
        _require_cuda(x, "training.conv2d_s2")
        B, H, W, Cp = x.shape
        Co, Ci = w.shape[0], w.shape[1]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dev = x.device
        K = _rup(9 * Cp, 32)
        rows = B * Ho * Wo
        col = (torch.zeros if K > 9 * Cp else torch.empty)(rows, K, device=dev, dtype=torch.float32)
        _call("srn_im2col_s2", x, col, B, H, W, Cp, K)
        wk = torch.zeros(Co, K, device=dev, dtype=torch.float32)  # [co][(kh, kw, ci)]
        wk[:, :9 * Cp].view(Co, 3, 3, Cp)[..., :Ci] = w.permute(0, 2, 3, 1)
        y = torch.empty(B, Ho, Wo, Co, device=dev, dtype=torch.float32)
        _launch_conv(col, wk, None, y, [0], 1, rows, rows, K, Co)
        ctx.save_for_backward(col, wk)
        ctx.dims = (B, H, W, Cp, Co, Ci, Ho, Wo, K, rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        col, wk = ctx.saved_tensors
        B, H, W, Cp, Co, Ci, Ho, Wo, K, rows = ctx.dims
        dev = dy.device
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wd = torch.empty(K, Co, device=dev, dtype=torch.float32)
            _call("srn_transpose_ct", wk, wd, 1, Co, K, 0, K, 0, Co)
            dcol = torch.empty(rows, K, device=dev, dtype=torch.float32)
            _launch_conv(dy, wd, None, dcol, [0], 1, rows, rows, Co, K)
            dx = torch.empty(B, H, W, Cp, device=dev, dtype=torch.float32)
            _call("srn_col2im_s2", dcol, dx, B, H, W, Cp, K)
        if ctx.needs_input_grad[1]:
            dwk = torch.empty(Co, K, device=dev, dtype=torch.float32)
            TnGemmOp(a=dy, b=col, out=dwk, n_items=1, T_a=rows, T_b=rows, M=Co, N=K, lda=Co, ldb=K, ldc=K)()
            dw = dwk[:, :9 * Cp].view(Co, 3, 3, Cp)[..., :Ci].permute(0, 3, 1, 2)  # -> (Co, Ci, kh, kw)
        return dx, dw


def conv2d_s2(x, w):
    return _Conv2dS2.apply(x.contiguous(), w)


class _BnRelu(torch.autograd.Function):
    """relu(BatchNorm(x)) over the rows of channels-last x (..., C) with batch statistics; running statistics (when
    given) move like nn.BatchNorm2d's (momentum 0.1)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, eps, momentum):
        _require_cuda(x, "training.bn_relu")
        C = x.shape[-1]
        rows = x.numel() // C
        x = x.contiguous()
        gamma, beta = gamma.contiguous(), beta.contiguous()
        dev = x.device
        part = torch.empty(_bn_chunks(rows) * 2 * C, device=dev, dtype=torch.float32)
        stats = torch.empty(2, C, device=dev, dtype=torch.float32)
        y = torch.empty_like(x)
        _call("srn_bn_relu_fwd", x, gamma, beta, run_mean, run_var, part, stats, y, rows, C, eps, momentum)
        ctx.save_for_backward(x, y, stats, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        dev = dy.device
        part = torch.empty(_bn_chunks(rows) * 2 * C, device=dev, dtype=torch.float32)
        sums = torch.empty(2, C, device=dev, dtype=torch.float32)
        dx = torch.empty_like(x)
        _call("srn_bn_relu_bwd", x, y, dy.contiguous(), stats, gamma, part, sums, dx, rows, C)
        return dx, sums[1], sums[0], None, None, None, None


def _bn_chunks(rows):
    per = 16 if rows <= 256 * 16 else (rows + 255) // 256  # srn_bn_chunks
    return (rows + per - 1) // per


def bn_relu(x, gamma, beta, run_mean=None, run_var=None, eps=1e-5, momentum=0.1):
    return _BnRelu.apply(x, gamma, beta, run_mean, run_var, eps, momentum)


class _GruLast(torch.autograd.Function):
    """last hidden state of nn.GRU (one layer, zero initial state) given gi = x W_ih^T + b_ih (B, T, 3H)."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh):
        _require_cuda(gi, "training.gru_last")
        B, T, G = gi.shape
        H = G // 3
        dev = gi.device
        gi, w_hh = gi.contiguous(), w_hh.contiguous()
        w_hh_t = torch.empty(H, G, device=dev, dtype=torch.float32)
        _call("srn_transpose_ct", w_hh, w_hh_t, 1, G, H, 0, H, 0, G)
        hs = torch.empty(B, T + 1, H, device=dev, dtype=torch.float32)
        gates = torch.empty(B, T, 4 * H, device=dev, dtype=torch.float32)
        _call("srn_gru_train_fwd", gi, w_hh_t, b_hh.contiguous(), hs, gates, B, T, H)
        ctx.save_for_backward(w_hh, hs, gates)
        return hs[:, T].clone()

    @staticmethod
    def backward(ctx, dh):
        w_hh, hs, gates = ctx.saved_tensors
        B, T1, H = hs.shape
        T, G = T1 - 1, 3 * H
        dev = dh.device
        dgi = torch.empty(B, T, G, device=dev, dtype=torch.float32)
        dgh = torch.empty(B, T, G, device=dev, dtype=torch.float32)
        _call("srn_gru_train_bwd", dh.contiguous(), w_hh, hs, gates, dgi, dgh, B, T, H)
        dw = torch.empty(G, H, device=dev, dtype=torch.float32)
        TnGemmOp(a=dgh, b=hs, out=dw, n_items=B, T_a=T, T_b=T, M=G, N=H, lda=G, ldb=H, ldc=H, a_is=T * G,
                 b_is=(T + 1) * H)()
        return dgi, dw, _colsum(dgh.view(B * T, G))


def gru_last(gi, w_hh, b_hh):
    return _GruLast.apply(gi, w_hh, b_hh)


class _TokenAttn(torch.autograd.Function):
    """gst/attention.py:110-184 for one query per item: q (B, F), k / v (n_tok, F) -> context (B, F)."""

    @staticmethod
    def forward(ctx, q, k, v, n_head):
        _require_cuda(q, "training.token_attention")
        B, Fd = q.shape
        n_tok = k.shape[0]
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        p = torch.empty(B, n_head, n_tok, device=q.device, dtype=torch.float32)
        out = torch.empty(B, Fd, device=q.device, dtype=torch.float32)
        _call("srn_token_attn_fwd", q, k, v, p, out, B, n_tok, Fd, n_head)
        ctx.save_for_backward(q, k, v, p)
        ctx.n_head = n_head
        return out

    @staticmethod
    def backward(ctx, dctx):
        q, k, v, p = ctx.saved_tensors
        B, Fd = q.shape
        n_tok = k.shape[0]
        dev = q.device
        dq = torch.empty(B, Fd, device=dev, dtype=torch.float32)
        dkp = torch.empty(B, n_tok * Fd, device=dev, dtype=torch.float32)
        dvp = torch.empty(B, n_tok * Fd, device=dev, dtype=torch.float32)
        _call("srn_token_attn_bwd", dctx.contiguous(), q, k, v, p, dq, dkp, dvp, B, n_tok, Fd, ctx.n_head)
        return dq, _colsum(dkp).view(n_tok, Fd), _colsum(dvp).view(n_tok, Fd), None


def token_attention(q, k, v, n_head):
    return _TokenAttn.apply(q, k, v, n_head)


# =====================================================================================================================
#  GEGLU
# =====================================================================================================================
class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hg):
        rows, two = hg.numel() // hg.shape[-1], hg.shape[-1]
        a = torch.empty(*hg.shape[:-1], two // 2, device=hg.device, dtype=torch.float32)
        _call("srn_geglu_fwd", hg, a, rows, two // 2)
        ctx.save_for_backward(hg)
        return a

    @staticmethod
    def backward(ctx, da):
        (hg,) = ctx.saved_tensors
        rows, two = hg.numel() // hg.shape[-1], hg.shape[-1]
        dhg = torch.empty_like(hg)
        _call("srn_geglu_bwd", hg, da.contiguous(), dhg, rows, two // 2)
        return dhg


def geglu(hg):
    """(.., 2 inner) = [h | g] -> h * gelu_erf(g)   (transformer.py:120-146)"""
    return _Geglu.apply(hg.contiguous())


# =====================================================================================================================
#  the estimator
# =====================================================================================================================
def sinusoidal_pos_emb(t, dim, scale=1000.0):
    """decoder.py:54-63 (no parameters): t (B,) -> (B, dim)"""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, device=t.device, dtype=torch.float32) * -e)
    arg = scale * t.reshape(-1, 1).to(torch.float32) * f.unsqueeze(0)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


class ParamStore:
    """Trainable tensors under their state_dict names, every one a view of ONE flat fp32 buffer (`flat`) with its
    gradient a view of `flat_grad` (autograd accumulates in place): what the all-reduce and the optimizer work on.
    Tensors in `skip` (BatchNorm running statistics) are not parameters and stay out."""

    def __init__(self, state_dict, device, skip=()):
        names = [k for k in state_dict if k not in skip]
        sizes = [state_dict[k].numel() for k in names]
        offs, o = [], 0
        for n in sizes:  # 4-float alignment of every view (16-B loads in the kernels)
            offs.append(o)
            o += _rup(n, 4)
        self.device = device
        self.flat = torch.zeros(o, device=device, dtype=torch.float32)
        self.flat_grad = torch.zeros(o, device=device, dtype=torch.float32)
        self.params, self.spans = {}, {}
        for k, off, n in zip(names, offs, sizes):
            v = self.flat[off:off + n].view(state_dict[k].shape)
            v.copy_(state_dict[k].to(torch.float32))
            v.requires_grad_(True)
            v.grad = self.flat_grad[off:off + n].view(state_dict[k].shape)
            self.params[k] = v
            self.spans[k] = (off, n)

    def zero_grad(self):
        self.flat_grad.zero_()

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.params.items()}

    def backward_into_flat(self, loss):
        """`flat_grad` <- d loss / d parameters, OVERWRITING it (no zero_grad needed): torch.autograd.grad + one
        multi-tensor copy (srn_multi_copy), every op on the current stream.  Used by captured steps: the per-leaf AccumulateGrad nodes of
        `loss.backward()` run on the stream the leaf was created on, and the 262 in-place adds they issue from there
        raced with the capture stream's reuse of the gradient temporaries."""
        names = list(self.params)
        ps = [self.params[k] for k in names]
        grads = torch.autograd.grad(loss, ps, allow_unused=True)
        with torch.no_grad():
            have = [(g.contiguous(), self.spans[k][0]) for k, g in zip(names, grads) if g is not None]
            # one launch per 160 tensors (torch._foreach_copy_ issues one memcpy per tensor on this stack: 245 launches)
            ops.MultiCopyOp([g for g, _ in have], [o for _, o in have], self.flat_grad)()
            for p, g in zip(ps, grads):
                if g is None:
                    p.grad.zero_()


class Estimator:
    """`Decoder` (matcha_components/decoder.py:196-467) for training: parameters under the reference's names and
    shapes (`self.params`), each a view of the flat buffer `self.flat`; gradients are views of `self.flat_grad`.

    `forward(x, mask, mu, t, spks)` takes the reference's layouts -- x (B, out_ch, L), mask (B, 1, L), mu (B, cond, L),
    t (B,), spks (B, S) -- and returns (B, out_ch, L) with an autograd graph whose backward runs on the HIP kernels."""

    N_HEAD, GROUPS = 4, 8

    def __init__(self, state_dict, device, dropout=0.0, _store=None, _prefix=""):
        """state_dict: the estimator's tensors under the reference's names.  dropout: probability of the two Dropout
        layers of every transformer block (attention output, feed-forward; decoder.py:215 trains with 0.05) -- torch's
        generator, so not reproducible against the reference's draws; parity tests use 0."""
        if _store is None:
            _store = ParamStore(state_dict, device)
        self.store = _store
        self.flat, self.flat_grad, self.device = _store.flat, _store.flat_grad, _store.device
        n = len(_prefix)
        self.params = {k[n:]: v for k, v in _store.params.items() if k.startswith(_prefix)}
        self.spans = {k[n:]: v for k, v in _store.spans.items() if k.startswith(_prefix)}
        self.dropout = float(dropout)
        names = list(self.params)
        self.n_down = sum(1 for k in names if k.startswith("down_blocks.") and k.endswith(".0.mlp.1.weight"))
        self.n_mid = sum(1 for k in names if k.startswith("mid_blocks.") and k.endswith(".0.mlp.1.weight"))
        self.n_up = sum(1 for k in names if k.startswith("up_blocks.") and k.endswith(".0.mlp.1.weight"))
        self._plan_relay()

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.params.items()}

    def zero_grad(self):
        self.flat_grad.zero_()

    def _plan_relay(self):
        """What every step re-lays of the weights, planned once: conv weights (N, C, k) -> packed (N, k * C_pad)
        (`self._pk`, the differentiable outputs of _PackAll) and, for every conv / linear, W^T for the input gradient
        (`self._wd[name]`: (C_pad, k * N); q | k | v side by side for the fused projection).  Buffers are allocated here
        (pad columns stay zero) and the transposes are two prebuilt srn_transpose_multi tables: parameters first, then
        the transposes that read the packed buffers."""
        P, dev = self.params, self.device
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self._pk, self._wd, self._pack_names, self._relay_ops = {}, {}, [], []
        if "down_blocks.0.0.block1.block.0.weight" not in P:
            return  # not a decoder's state_dict (GradSync's unit tests hand a bare ParamStore): nothing to re-lay
        cin0 = P["down_blocks.0.0.block1.block.0.weight"].shape[1]
        first, second = [], []
        for name, w in P.items():
            if not name.endswith(".weight") or w.dim() not in (2, 3):
                continue
            if name.startswith("up_blocks.") and name.endswith("2.conv.weight"):
                continue  # ConvTranspose1d: re-laid per output phase by _convtranspose_phases
            if name.startswith("time_mlp.linear_1"):
                continue  # its input is padded on the fly (242 -> 244 columns)
            if any(t in name for t in (".to_q.", ".to_k.", ".to_v.")):
                continue  # the fused projection, below
            if w.dim() == 2 and w.shape[0] * w.shape[1] >= 64:
                n, c = w.shape
                if ".norm" in name:
                    continue
                self._wd[name] = z(c, n)
                first.append((w, self._wd[name], 1, n, c, 0, c, 0, n))
            elif w.dim() == 3:
                n, c, k = w.shape
                c_pad = _rup(c, 32) if name.startswith("down_blocks.0.0.") and c == cin0 else c
                self._wd[name] = z(c_pad, k * n)
                if k == 1 and c_pad == c:  # pack_conv is a view
                    first.append((w, self._wd[name], 1, n, c, 0, c, 0, n))
                    continue
                self._pk[name] = z(n, k * c_pad)
                self._pack_names.append(name)
                first.append((w, self._pk[name], n, c, k, c * k, k, k * c_pad, c_pad))
                second.append((self._pk[name], self._wd[name], k, n, c_pad, c_pad, k * c_pad, n, k * n))
        for name in [k for k in P if k.endswith("attn1.to_q.weight")]:
            pre = name[:-len("to_q.weight")]
            ws = [P[pre + t + ".weight"] for t in ("to_q", "to_k", "to_v")]
            inner, c = ws[0].shape
            wd = z(c, 3 * inner)
            self._wd[pre + "qkv"] = wd
            for i, w in enumerate(ws):
                first.append((w, (wd, i * inner), 1, inner, c, 0, c, 0, 3 * inner))
        self._relay_ops = [ops.TransposeMultiOp(first), ops.TransposeMultiOp(second)]

    def _relay(self):
        """once per forward: packed conv weights (differentiable) + the input-gradient layouts (constants of the step)"""
        packed = _PackAll.apply(self, *[self.params[n] for n in self._pack_names])
        self._pkd = dict(zip(self._pack_names, packed))

    def _packed(self, name, c_pad=None):
        w = self._pkd.get(name)
        return w if w is not None else pack_conv(self.params[name], c_pad)

    def _drop(self, x):
        return F.dropout(x, self.dropout, True) if self.dropout > 0.0 else x

    # ---- blocks ------------------------------------------------------------------------------------------------
    def _lin(self, x, name, bias=True, c_pad=None):
        w = self.params[name + ".weight"]
        if c_pad is not None and c_pad > w.shape[1]:
            w = F.pad(w, (0, c_pad - w.shape[1]))
        wd = self._wd.get(name + ".weight") if c_pad is None else None
        return conv1d(x, w, self.params[name + ".bias"] if bias else None, wd=wd)

    def _block1d(self, p, x, maskf, lens, c_pad=None):
        """Block1D (decoder.py:66-77): conv k3 of the masked input -> GroupNorm(8) -> Mish -> mask"""
        w = self._packed(p + "block.0.weight", c_pad)
        h, part = conv1d(x, w, self.params[p + "block.0.bias"], ops.conv_taps(3), want_gn=True, lens=lens,
                         wd=self._wd.get(p + "block.0.weight"))
        return gn_mish(h, part, self.params[p + "block.1.weight"], self.params[p + "block.1.bias"], lens, self.GROUPS)

    def _resnet(self, p, x, maskf, lens, temb, spk, c_pad=None):
        """ResnetBlock1D (decoder.py:80-101) + SpeakerAdapter (decoder.py:23-45)"""
        h = self._block1d(p + "block1.", x, maskf, lens, c_pad)
        h = _AddRowBias.apply(h, self._lin(F.mish(temb), p + "mlp.1"))
        h = self._block1d(p + "block2.", h, maskf, lens)
        out = h + conv1d(x, self._packed(p + "res_conv.weight", c_pad), self.params[p + "res_conv.bias"], lens=lens,
                         wd=self._wd.get(p + "res_conv.weight"))
        scale = self._lin(spk, p + "speaker_projection.W_scale")
        shift = self._lin(spk, p + "speaker_projection.W_bias")
        return row_ln(out, scale, shift)

    def _tfm(self, p, x, lens):
        """BasicTransformerBlock effective path (transformer.py:286-352): LN -> self-attention -> +x -> LN -> GEGLU FF
        -> +x; dropout not applied (see module docstring)"""
        P = self.params
        n = row_ln(x, P[p + "norm1.weight"], P[p + "norm1.bias"])
        wqkv = torch.cat([P[p + "attn1.to_q.weight"], P[p + "attn1.to_k.weight"], P[p + "attn1.to_v.weight"]], dim=0)
        o = attention_core(conv1d(n, wqkv, None, wd=self._wd.get(p + "attn1.qkv")), lens, self.N_HEAD)
        x = self._drop(self._lin(o, p + "attn1.to_out.0")) + x
        n = row_ln(x, P[p + "norm3.weight"], P[p + "norm3.bias"])
        a = self._drop(geglu(self._lin(n, p + "ff.net.0.proj")))
        return self._lin(a, p + "ff.net.2") + x

    def forward(self, x, mask, mu, t, spks):
        _require_cuda(x, "Estimator.forward")
        P = self.params
        self._relay()
        B, _, L = x.shape
        maskb = mask.reshape(B, L) > 0
        if not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
            # the reference's `x * mask` is done by row counts here (len_in / len_out / len_b of the kernels): that equals
            # the multiply only for a LENGTH mask -- anything else would silently change activations and gradients
            if not torch.equal(maskb, torch.arange(L, device=x.device)[None] < maskb.sum(1, keepdim=True)):
                raise ValueError("Estimator.forward: `mask` must be a length (prefix) mask, as make_non_pad_mask builds it")
        h = torch.cat([x, mu], dim=1).transpose(1, 2)  # (B, L, 242) channels-last
        cin = h.shape[-1]
        cp = _rup(cin, 32)
        h = F.pad(h, (0, cp - cin)).contiguous()
        t = torch.as_tensor(t, device=x.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        s = sinusoidal_pos_emb(t, cin)
        s = F.pad(s, (0, _rup(cin, 4) - cin))
        temb = self._lin(F.silu(self._lin(s, "time_mlp.linear_1", c_pad=s.shape[1])), "time_mlp.linear_2")
        spk = spks.to(torch.float32)

        masks = [maskb]
        hiddens = []
        for i in range(self.n_down):
            p = f"down_blocks.{i}."
            m = masks[-1]
            lens, mf = m.sum(1).to(torch.int32), m.unsqueeze(-1).to(torch.float32)
            h = self._resnet(p + "0.", h, mf, lens, temb, spk, c_pad=cp if i == 0 else None)
            h = self._tfm(p + "1.0.", h, lens)
            hiddens.append(h)
            if p + "2.conv.weight" in P:
                h = conv1d(h * mf, self._packed(p + "2.conv.weight"), P[p + "2.conv.bias"], ops.conv_taps(3), stride=2,
                           wd=self._wd.get(p + "2.conv.weight"))
            else:
                h = conv1d(h, self._packed(p + "2.weight"), P[p + "2.bias"], ops.conv_taps(3), lens=lens,
                           wd=self._wd.get(p + "2.weight"))
            masks.append(m[:, ::2])
        masks = masks[:-1]
        m = masks[-1]
        lens, mf = m.sum(1).to(torch.int32), m.unsqueeze(-1).to(torch.float32)
        for i in range(self.n_mid):
            p = f"mid_blocks.{i}."
            h = self._resnet(p + "0.", h, mf, lens, temb, spk)
            h = self._tfm(p + "1.0.", h, lens)
        for i in range(self.n_up):
            p = f"up_blocks.{i}."
            m = masks.pop()
            lens, mf = m.sum(1).to(torch.int32), m.unsqueeze(-1).to(torch.float32)
            skip = hiddens.pop()
            h = torch.cat([h[:, :skip.shape[1]], skip], dim=-1)
            h = self._resnet(p + "0.", h, mf, lens, temb, spk)
            h = self._tfm(p + "1.0.", h, lens)
            if p + "2.conv.weight" in P:
                # ConvTranspose1d(k 4, stride 2, padding 1): two output phases, each a 2-tap conv of the input
                ys = []
                for taps, wp in _convtranspose_phases(P[p + "2.conv.weight"], 2, 1):
                    ys.append(conv1d(h, wp, P[p + "2.conv.bias"], taps, lens=lens))
                h = torch.stack(ys, dim=2).reshape(B, -1, ys[0].shape[-1])
            else:
                h = conv1d(h, self._packed(p + "2.weight"), P[p + "2.bias"], ops.conv_taps(3), lens=lens,
                           wd=self._wd.get(p + "2.weight"))
        h = self._block1d("final_block.", h, mf, lens)
        out = conv1d(h, self._packed("final_proj.weight"), P["final_proj.bias"], lens=lens,
                     wd=self._wd.get("final_proj.weight"))
        return (out * maskb.unsqueeze(-1).to(torch.float32)).transpose(1, 2)

    __call__ = forward


def _convtranspose_phases(w, stride, padding):
    """differentiable twin of ops.convtranspose_phases: ConvTranspose1d weight (Ci, Co, k) -> per output phase
    (input-row offsets, packed (Co, n_taps * Ci))"""
    ci, co, k = w.shape
    out = []
    for r in range(stride):
        taps, mats = [], []
        j = -((r + padding) // stride)
        while True:
            kk = r + padding + j * stride
            if kk >= k:
                break
            if kk >= 0:
                taps.append(-j)
                mats.append(w[:, :, kk].t())
            j += 1
        out.append((taps, torch.stack(mats, dim=1).reshape(co, len(taps) * ci)))
    return out


class _Dot(torch.autograd.Function):
    """sum(a * b) (b None: sum(a)) through srn_dot's fp64 partial sums.  Not a torch reduction: one of this size zeroes
    its semaphore buffer with hipMemsetAsync, and a captured memset node replays with a corrupted value on this stack
    (count_memset_nodes); the <= 1024 partials are added by a single-block torch sum, which has no such node."""

    @staticmethod
    def forward(ctx, a, b):
        a = a.contiguous().view(-1)
        b = None if b is None else b.contiguous().view(-1)
        if a.data_ptr() % 16:  # srn_dot reads 16-byte pieces: a sliced view may start anywhere
            a = a.clone()
        if b is not None and b.data_ptr() % 16:
            b = b.clone()
        part = torch.zeros(1024, device=a.device, dtype=torch.float64)
        _call("srn_dot", a, b, a.numel(), part)
        ctx.save_for_backward(a, b)
        return part.sum().to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        if b is None:
            return g.expand_as(a), None
        return g * b, g * a


def _dot(a, b=None):
    return _Dot.apply(a.reshape(-1), None if b is None else b.reshape(-1))


def _total(x):
    """sum of all elements"""
    return _dot(x)


def cfm_loss(estimator, x1, mask, mu, spks, mask_l=None, draws=None, sigma_min=1e-4):
    """CFM.compute_loss (flow_matching.py:95-133) with autograd through `estimator`.  `draws` = {"t": (B,1,1),
    "z": like x1} fixes the random draws (parity tests); otherwise torch.rand / randn_like as in the reference."""
    b = mu.shape[0]
    if draws is None:
        t = torch.rand([b, 1, 1], device=mu.device, dtype=mu.dtype)
        z = torch.randn_like(x1)
    else:
        t, z = draws["t"].to(x1), draws["z"].to(x1)
    y = (1 - (1 - sigma_min) * t) * z + t * x1
    u = x1 - (1 - sigma_min) * z
    den = estimator(y, mask, mu, t.reshape(-1), spks)
    if mask_l is not None:
        den = den * mask_l
        u = u * mask_l
    d = (den - u).reshape(-1)
    loss = _dot(d, d)  # F.mse_loss(den, u, reduction="sum")
    denom = _total(mask_l if mask_l is not None else mask.to(d.dtype))
    return loss / (denom * u.shape[1]), y


# =====================================================================================================================
#  the whole model: Serenade.forward (serenade.py:90-166) for training
# =====================================================================================================================
_REFLECT_IDX = {}


def _reflect_pad_rows(x, p):
    """nn.ReflectionPad1d(p) along time of a channels-last (B, T, C) tensor (a gather: autograd scatters back)"""
    T = x.shape[1]
    d = x.device  # built on the device: a captured step may not copy from the host
    key = (T, p, str(d))
    idx = _REFLECT_IDX.get(key)
    if idx is None:  # once per (length, pad): three aranges + a cat per call otherwise, five calls per step
        idx = _REFLECT_IDX[key] = torch.cat([torch.arange(p, 0, -1, device=d), torch.arange(T, device=d),
                                             torch.arange(T - 2, T - 2 - p, -1, device=d)])
    return x.index_select(1, idx)


class TrainSerenade:
    """`Serenade` (serenade/models/serenade.py:35-166) for training: all trainable tensors of the checkpoint in one
    `ParamStore`; `forward` returns {"gauss_mel", "prior_loss", "cfm_loss"} with an autograd graph.

      * estimator: `Estimator` above (HIP forward and backward);
      * content encoder `Conv1dResnet` (serenade.py:282-296,310-376): weight-norm folded by torch ops, reflection
        padding as a row gather, every conv through `conv1d` (HIP forward, dgrad; rocBLAS wgrad), LeakyReLU in torch;
      * GST style encoder (modules/gst/style_encoder.py): six 3x3 stride-2 Conv2d + train-mode BatchNorm2d + ReLU, a
        GRU over T_ref / 64 steps and the 50-token attention -- 0.8 GFLOP of the step's ~1.3 TFLOP -- run on
        torch's GPU ops (MIOpen / rocBLAS) in both directions: the library path, not hand-written kernels.
    BatchNorm running statistics are updated like nn.BatchNorm2d (momentum 0.1) in `self.buffers`."""

    def __init__(self, state_dict, device, dropout=0.05, mask_size=(0.1, 0.5), output_dim=80):
        skip = [k for k in state_dict if k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
        self.store = ParamStore(state_dict, device, skip=skip)
        self.buffers = {k: state_dict[k].to(device).clone() for k in skip}
        self.flat, self.flat_grad, self.device = self.store.flat, self.store.flat_grad, device
        self.params, self.spans = self.store.params, self.store.spans
        self.estimator = Estimator(None, device, dropout=dropout, _store=self.store, _prefix="cfm_decoder.estimator.")
        self.mask_size, self.output_dim, self.training = tuple(mask_size), output_dim, True

    def zero_grad(self):
        self.store.zero_grad()

    def backward(self, loss, sync=None):
        """gradients of `loss` into the flat buffer, all-reduced if `sync` spans several ranks.  With a GradSync that
        overlaps: zero_grad() + `loss.backward()`, whose per-parameter hooks launch the bucketed all-reduce while
        backward is still running, + sync.finish().  Otherwise one `torch.autograd.grad` + one multi-tensor copy that
        OVERWRITES the buffer -- no zero fill, none of the 262 per-parameter accumulation kernels -- then sync.finish()."""
        if sync is not None and sync._hooks:
            self.zero_grad()
            loss.backward()
        else:
            self.store.backward_into_flat(loss)
        if sync is not None:
            sync.finish()

    def state_dict(self):
        sd = self.store.state_dict()
        sd.update({k: v.clone() for k, v in self.buffers.items()})
        return sd

    # ---- content encoder -----------------------------------------------------------------------------------------
    def _wn(self, name):
        """(packed weight, W^T for the input gradient or None) of an encoder conv: weight-normed ones through the fused
        kernel, plain ones through pack_conv"""
        P = self.params
        if name + ".weight" in P:
            return pack_conv(P[name + ".weight"]), None
        return _WeightNormPack.apply(P[name + ".weight_v"], P[name + ".weight_g"])

    def encoder(self, x):
        """(B, T, in_dim) -> (B, T, out_dim); like the reference, padded frames are not masked here"""
        P, T = self.params, x.shape[1]
        e = "encoder.model."
        w, wd = self._wn(e + "1")
        h = conv1d(_reflect_pad_rows(x.contiguous(), 3), w, P[e + "1.bias"], range(7), T_out=T, wd=wd)
        n = 0
        while f"{e}{2 + n}.shortcut.bias" in P:
            p, d = f"{e}{2 + n}", 2 ** n
            w, wd = self._wn(p + ".shortcut")
            sc = conv1d(h, w, P[p + ".shortcut.bias"], wd=wd)
            w, wd = self._wn(p + ".block.2")
            b = conv1d(_reflect_pad_rows(F.leaky_relu(h, 0.2), d), w, P[p + ".block.2.bias"], [0, d, 2 * d], T_out=T, wd=wd)
            w, wd = self._wn(p + ".block.4")
            b = conv1d(F.leaky_relu(b, 0.2), w, P[p + ".block.4.bias"], wd=wd)
            h = sc + b
            n += 1
        last = f"{e}{2 + n + 2}"
        w, wd = self._wn(last)
        return conv1d(_reflect_pad_rows(F.leaky_relu(h, 0.2), 3), w, P[last + ".bias"], range(7), T_out=T, wd=wd)

    # ---- GST -------------------------------------------------------------------------------------------------------
    def gst(self, speech, n_head=4):
        """StyleEncoder.forward (style_encoder.py:78-91,171-191,235-252), channels-last, own kernels in both directions:
        Conv2d(k3, s2, p1) as one stride-2 three-tap contraction along the mel axis per kernel row (srn_conv_gemm forward
        and dgrad, srn_tn_gemm wgrad), BatchNorm2d(training) + ReLU (srn_bn_relu_*), the GRU's input projection as a
        linear + srn_gru_train_* for the recurrence / BPTT, the token attention's four linears + srn_token_attn_*."""
        P, Bf = self.params, self.buffers
        r = "gst.ref_enc."
        B, T, F0 = speech.shape
        h = F.pad(speech.to(torch.float32).unsqueeze(-1), (0, 3))  # (B, H = T, W = 80, C = 1 padded to 4)
        i = 0
        while f"{r}convs.{3 * i}.weight" in P:
            c, b = f"{r}convs.{3 * i}", f"{r}convs.{3 * i + 1}"
            h = conv2d_s2(h, P[c + ".weight"])
            if self.training:
                h = bn_relu(h, P[b + ".weight"], P[b + ".bias"], Bf[b + ".running_mean"], Bf[b + ".running_var"])
                if b + ".num_batches_tracked" in Bf:
                    Bf[b + ".num_batches_tracked"] += 1
            else:
                # eval mode (the reference's _eval_epoch, trainers/base.py:171-190): BatchNorm2d on the running
                # statistics, folded into one scale / shift per channel (channels last), then ReLU
                scale = P[b + ".weight"] * torch.rsqrt(Bf[b + ".running_var"] + 1e-5)
                h = torch.relu(h * scale + (P[b + ".bias"] - Bf[b + ".running_mean"] * scale))
            i += 1
        bsz, tlen, wdim, cdim = h.shape
        xs = h.reshape(bsz, tlen, wdim * cdim)  # features ordered (w, c); the reference's view orders them (c, w)
        wih, whh = P[r + "gru.weight_ih_l0"], P[r + "gru.weight_hh_l0"]
        wih = wih.view(-1, cdim, wdim).permute(0, 2, 1).reshape(wih.shape[0], -1)
        gi = conv1d(xs, wih, P[r + "gru.bias_ih_l0"])
        hh = gru_last(gi, whh, P[r + "gru.bias_hh_l0"])
        m = "gst.stl.mha."
        toks = torch.tanh(P["gst.stl.gst_embs"])
        q = conv1d(hh, P[m + "linear_q.weight"], P[m + "linear_q.bias"])
        k = conv1d(F.pad(toks, (0, (-toks.shape[1]) % 4)), F.pad(P[m + "linear_k.weight"], (0, (-toks.shape[1]) % 4)),
                   P[m + "linear_k.bias"])
        v = conv1d(F.pad(toks, (0, (-toks.shape[1]) % 4)), F.pad(P[m + "linear_v.weight"], (0, (-toks.shape[1]) % 4)),
                   P[m + "linear_v.bias"])
        ctx = token_attention(q, k, v, n_head)
        return conv1d(ctx, P[m + "linear_out.weight"], P[m + "linear_out.bias"])

    def draw_segment(self, T):
        """(seg_start, seg_len) of the infill segment, drawn like serenade.py:117-119 (python `random`)"""
        import random
        msize = int(random.uniform(*self.mask_size) * T)
        return random.randint(0, T - msize), msize

    # ---- Serenade.forward ----------------------------------------------------------------------------------------
    def forward(self, x, lengths, logmel, midi, lft, draws=None):
        """serenade.py:90-166.  `draws` (tests) = {"uniform", "seg_start", "t", "z"} replaces the random draws."""
        import random
        _require_cuda(x, "TrainSerenade.forward")
        ret = {}
        enc = self.encoder(x.to(torch.float32))
        ret["gauss_mel"] = enc
        spk = self.gst(logmel)
        B, T = enc.shape[0], enc.shape[1]
        mask = (torch.arange(T, device=x.device)[None] < torch.as_tensor(lengths, device=x.device)[:, None])
        mask = mask.unsqueeze(1).to(torch.float32)
        if draws is not None and "seg" in draws:  # device tensor [seg_start, seg_len] (a captured step replays it)
            s0, msize = draws["seg"][0], draws["seg"][1]
        else:
            uni = random.uniform(*self.mask_size) if draws is None else float(draws["uniform"])
            msize = int(uni * T)
            s0 = random.randint(0, T - msize) if draws is None else int(draws["seg_start"])
        idx = torch.arange(T, device=x.device)
        inside = ((idx >= s0) & (idx < s0 + msize)).to(torch.float32).view(1, 1, T)
        mask_l = mask * inside
        mask_c = mask * (1.0 - inside)
        # sum(0.5 * ((logmel - enc)^2 + log 2 pi) * mask) over (B, out, T), written with dot products (see _total)
        diff = (logmel - enc).reshape(-1)
        n_valid = _total(mask)
        prior = 0.5 * _dot(diff * mask.permute(0, 2, 1).expand(B, T, self.output_dim).reshape(-1), diff) \
            + 0.5 * math.log(2 * math.pi) * self.output_dim * n_valid
        ret["prior_loss"] = prior / (n_valid * self.output_dim)
        targets = logmel * mask_l.permute(0, 2, 1)
        cond = logmel * mask_c.permute(0, 2, 1)
        mu = torch.cat([enc, midi, lft, cond], dim=-1)
        tz = None if draws is None or "t" not in draws else {"t": draws["t"], "z": draws["z"]}
        ret["cfm_loss"], _ = cfm_loss(self.estimator, targets.permute(0, 2, 1), mask, mu.permute(0, 2, 1), spk, mask_l,
                                      draws=tz)
        return ret

    __call__ = forward



# =====================================================================================================================
#  gradient all-reduce (DDP replacement) and the optimizer
# =====================================================================================================================
class GradSync:
    """All-reduce (mean) of `estimator.flat_grad` across the ranks, in buckets of ~`bucket_bytes`.

    The reference wraps the model in DistributedDataParallel over NCCL (bin/ssc_train.py:353-358).  Here the flat
    gradient buffer is cut into contiguous buckets; the parameters are laid out in forward order, so backward
    completes the buckets from the last to the first, and a post-accumulate hook on every parameter launches its
    bucket's asynchronous all-reduce the moment the bucket's last gradient has landed -- the RCCL transfers of the
    late layers overlap the backward GEMMs of the early ones.  xGMI rings are per-link bound, so buckets are large
    (default 64 MiB: 3 collectives for the 44 M-parameter estimator).  `finish()` waits and divides by the world size.
    World size 1 (or no process group) makes every call a no-op unless `always` (single-GPU rehearsal of the RCCL path)."""

    def __init__(self, estimator, bucket_bytes=64 << 20, group=None, always=False, overlap=True):
        import torch.distributed as dist
        self.dist = dist
        self.est = estimator
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.always = bool(always) and dist.is_available() and dist.is_initialized()  # collectives even with one rank
        self.launched = 0  # collectives issued so far (tests / logs)
        per = max(1, bucket_bytes // 4)
        total = estimator.flat.numel()
        self.buckets = []  # (start, end)
        lo = 0
        names = list(estimator.spans)
        for k in names:
            off, n = estimator.spans[k]
            end = off + _rup(n, 4)
            if end - lo >= per or k == names[-1]:
                self.buckets.append((lo, total if k == names[-1] else end))
                lo = end
        self.bucket_of = {}
        self.count = [0] * len(self.buckets)
        for k in names:
            off, _ = estimator.spans[k]
            bi = next(i for i, (s, e) in enumerate(self.buckets) if s <= off < e)
            self.bucket_of[k] = bi
            self.count[bi] += 1
        self._left = list(self.count)
        self._work = []
        self._hooks = []
        if (self.world > 1 or self.always) and overlap:  # overlap=False: everything is launched by finish()
            for k, p in estimator.params.items():
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(self.bucket_of[k])))

    def _make_hook(self, bi):
        def hook(_):
            self._left[bi] -= 1
            if self._left[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        s, e = self.buckets[bi]
        self.launched += 1
        self._work.append(self.dist.all_reduce(self.est.flat_grad[s:e], op=self.dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))

    def finish(self):
        """call after backward(): launches whatever the hooks did not (unused parameters), waits, averages"""
        if self.world == 1 and not self.always:
            return
        for bi, left in enumerate(self._left):
            if left > 0:
                self._launch(bi)
        for w in self._work:
            w.wait()
        self._work = []
        self._left = list(self.count)
        if self.world > 1:
            self.est.flat_grad.div_(self.world)


class AdamW:
    """clip_grad_norm_(max_norm) + torch.optim.AdamW (trainers/ssc.py:90-95; conf/serenade.yaml:62-65: lr 8e-4,
    grad_norm 1.0, torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0.01) as ONE kernel over the flat buffers."""

    def __init__(self, estimator, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0):
        self.est = estimator
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.m = torch.zeros_like(estimator.flat)
        self.v = torch.zeros_like(estimator.flat)
        self.steps = 0
        self.dyn = torch.zeros(4, device=estimator.flat.device, dtype=torch.float32)  # lr, bc1, bc2, grad_scale
        self._ss = torch.zeros(1024, device=estimator.flat.device, dtype=torch.float64)  # srn_sumsq partial sums
        self.norm = torch.zeros((), device=estimator.flat.device, dtype=torch.float64)  # last gradient norm (device)

    def _grad_norm(self):
        """fp64 total gradient norm as a 0-dim device tensor (own kernel + a 1024-element sum: torch's multi-block
        reductions gave wrong results when replayed inside a hipGraph on this stack)"""
        g = self.est.flat_grad
        _call("srn_sumsq", g, g.numel(), self._ss)
        return torch.sqrt(self._ss.sum())

    def prepare(self):
        """host side of `step_captured`: advance the step count and upload {lr, 1 - beta1^t, 1 - beta2^t} (call before
        every replay of a graph that contains `step_captured`)"""
        self.steps += 1
        h = torch.tensor([self.lr, 1.0 - self.betas[0] ** self.steps, 1.0 - self.betas[1] ** self.steps],
                         dtype=torch.float32)
        self.dyn[:3].copy_(h)

    def step_captured(self):
        """the same update with no host synchronisation (gradient norm, clip factor and bias corrections stay on the
        device): identical launches every step, so it can sit inside a captured hipGraph.  `prepare()` first."""
        g = self.est.flat_grad
        with torch.no_grad():
            norm = self._grad_norm()
            self.norm.copy_(norm)
            if self.max_norm and self.max_norm > 0:
                self.dyn[3:4] = torch.clamp(self.max_norm / (norm + 1e-6), max=1.0).to(torch.float32)
            else:
                self.dyn[3:4] = 1.0
            _call("srn_adamw_dyn", self.est.flat, g, self.m, self.v, g.numel(), self.betas[0], self.betas[1], self.eps,
                  self.wd, self.dyn)

    def step(self):
        """returns the gradient norm before clipping (what clip_grad_norm_ returns)"""
        _require_cuda(self.est.flat, "AdamW.step")
        g = self.est.flat_grad
        norm = float(self._grad_norm())
        scale = 1.0
        if self.max_norm and self.max_norm > 0:
            scale = min(1.0, self.max_norm / (norm + 1e-6))
        self.steps += 1
        with torch.no_grad():
            _call("srn_adamw", self.est.flat, g, self.m, self.v, g.numel(), self.lr, self.betas[0], self.betas[1],
                  self.eps, self.wd, self.steps, scale)
        return norm


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR for `AdamW` above (conf/serenade.yaml:66-70: gamma 0.5 at step 100 000):
    call step() once per optimizer step, as trainers/ssc.py:96 does."""

    def __init__(self, opt, milestones, gamma=0.5):
        self.opt, self.milestones, self.gamma = opt, sorted(int(m) for m in milestones), float(gamma)
        self.base_lr, self.last_epoch = opt.lr, 0

    def step(self):
        self.last_epoch += 1
        self.opt.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)

    def state_dict(self):
        """the entries of torch.optim.lr_scheduler.MultiStepLR.state_dict()"""
        from collections import Counter
        return {"milestones": Counter(self.milestones), "gamma": self.gamma, "base_lrs": [self.base_lr],
                "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1,
                "_get_lr_called_within_step": False, "_last_lr": [self.opt.lr]}

    def load_state_dict(self, sd):
        ms = sd["milestones"]
        self.milestones = sorted(int(m) for m in (ms.elements() if hasattr(ms, "elements") else ms))
        self.gamma, self.last_epoch = float(sd["gamma"]), int(sd["last_epoch"])
        self.base_lr = float(sd["base_lrs"][0] if "base_lrs" in sd else sd["base_lr"])
        self.opt.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)


def _torch_adamw_state(model, opt):
    """this optimizer's flat moments as `torch.optim.AdamW.state_dict()` of the reference's optimizer: parameters are
    numbered in `model.parameters()` order, which is the order of the parameter entries of the state_dict the store
    was built from (bin/ssc_train.py:331-343 builds AdamW over model.parameters())."""
    state, ids = {}, []
    for i, (k, (off, n)) in enumerate(model.spans.items()):
        shape = model.params[k].shape
        ids.append(i)
        if opt.steps > 0:
            state[i] = {"step": torch.tensor(float(opt.steps)), "exp_avg": opt.m[off:off + n].view(shape).cpu().clone(),
                        "exp_avg_sq": opt.v[off:off + n].view(shape).cpu().clone()}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": opt.wd, "amsgrad": False,
             "foreach": None, "maximize": False, "capturable": False, "differentiable": False, "fused": None,
             "params": ids}
    return {"state": state, "param_groups": [group]}


def _load_torch_adamw_state(o, model, opt):
    names = list(model.spans)
    groups = o["param_groups"]
    ids = [i for g in groups for i in g["params"]]
    if len(ids) != len(names):
        raise ValueError(f"optimizer state holds {len(ids)} parameters, this model has {len(names)}")
    opt.m.zero_(), opt.v.zero_()
    steps = 0
    for pos, i in enumerate(ids):
        st = o["state"].get(i)
        if st is None:
            continue
        off, n = model.spans[names[pos]]
        if st["exp_avg"].numel() != n:
            raise ValueError(f"optimizer state {i} has {st['exp_avg'].numel()} elements, {names[pos]} has {n}")
        opt.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
        opt.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps = max(steps, int(float(st["step"])))
    opt.steps, opt.lr = steps, float(groups[0]["lr"])


def save_checkpoint(path, model, opt, scheduler=None, steps=0, epochs=0):
    """the reference's checkpoint file (trainers/base.py:91-111): {"model", "optimizer", "scheduler", "steps",
    "epochs"} with "optimizer" / "scheduler" in the layout of `torch.optim.AdamW.state_dict()` /
    `MultiStepLR.state_dict()`, so the reference's trainer resumes from it and vice versa."""
    import os
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"model": {k: v.cpu() for k, v in model.state_dict().items()},
                "optimizer": _torch_adamw_state(model, opt),
                "scheduler": None if scheduler is None else scheduler.state_dict(), "steps": int(steps),
                "epochs": int(epochs)}, path)


def load_checkpoint(path, model, opt=None, scheduler=None, load_only_params=False):
    """trainers/base.py:113-130: restores the weights (and BatchNorm statistics) in place; with an optimizer also its
    moments, step count and learning rate -- from a checkpoint of this package or of the reference's trainer (torch's
    AdamW / MultiStepLR state_dicts).  Returns (steps, epochs)."""
    ck = torch.load(path, map_location="cpu")
    with torch.no_grad():
        for k, v in ck["model"].items():
            if k in model.params:
                model.params[k].copy_(v)
            elif k in getattr(model, "buffers", {}):
                model.buffers[k].copy_(v)
            else:
                raise KeyError(f"checkpoint tensor {k} is not a tensor of this model")
    if load_only_params or opt is None:
        return int(ck.get("steps", 0)), int(ck.get("epochs", 0))
    o = ck["optimizer"]
    if "param_groups" in o and "state" in o:
        with torch.no_grad():
            _load_torch_adamw_state(o, model, opt)
    elif "spans" in o:  # files written by round 2 of this package: flat moment buffers + their layout
        if dict(o["spans"]) != dict(model.spans):
            raise ValueError("optimizer state was saved for a different parameter layout")
        opt.m.copy_(o["m"]), opt.v.copy_(o["v"])
        opt.steps, opt.lr = int(o["steps"]), float(o["lr"])
    else:
        raise ValueError(f"unrecognised optimizer entry (keys {sorted(o)}): expected torch.optim.AdamW.state_dict()")
    if scheduler is not None and ck.get("scheduler") is not None:
        scheduler.load_state_dict(ck["scheduler"])
    return int(ck["steps"]), int(ck["epochs"])


def count_memset_nodes(graph):
    """number of memset nodes in a captured `torch.cuda.CUDAGraph(keep_graph=True)` (hipGraphGetNodes /
    hipGraphNodeGetType on its raw handle).

    Why it matters (profiles/r3_graph_probe.json, tools/experiments/graph_probe.py): on this stack (ROCm 7.2, torch
    2.10+rocm7.0) a hipMemsetAsync captured into a hipGraph fills with the right value on the FIRST replay and with
    garbage from the second replay on, for every size and value tried.  torch's multi-block reductions zero their
    semaphore buffer with exactly that call, so a captured `x.sum()` / `x.norm()` / `x.sum(0)` elects no last block
    and returns stale numbers from replay 2 on (forty reductions in one capture: error 0 on replay 1, 3e17 after).
    Kernels are unaffected.  Anything captured here must therefore hold no memset node."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    if hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    memsets = 0
    for i in range(n.value):
        kind = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(kind)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        memsets += kind.value == 2  # hipGraphNodeTypeMemset
    return memsets, n.value


class GraphedStep:
    """One whole-model training step of a fixed (B, L) captured as a hipGraph and replayed.

    At the reference's per-GPU batch (4) the eager step is bound by its ~2500 host-side launches; captured, the same
    kernels run back to back.  Everything step-dependent lives in device memory: the inputs (static buffers the call
    copies into), the infill segment (drawn on the host like the reference, uploaded as two integers), t / z / dropout
    (the device generator, graph-safe), the clip factor and Adam's bias corrections (`AdamW.step_captured`).
    With more than one rank the gradient all-reduce runs between two captured halves (backward | optimizer) on the whole
    flat buffer.  Training batches of varying length need one GraphedStep per length bucket."""

    def __init__(self, model, opt, B, L, in_dim=768, sync=None, warmup=2, tz=None):
        """tz (tests): (t (B,1,1), z (B,out,L)) device tensors used instead of the generator's draws"""
        dev = model.device
        self.model, self.opt, self.sync, self.tz = model, opt, sync, tz
        self.split = sync is not None and (sync.world > 1 or sync.always)
        if self.split and sync._hooks:
            raise ValueError("GraphedStep needs GradSync(..., overlap=False): collectives cannot sit inside the capture")
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.x, self.logmel, self.midi, self.lft = f(B, L, in_dim), f(B, L, model.output_dim), f(B, L, 1), f(B, L, 1)
        self.lens = torch.full((B,), L, device=dev, dtype=torch.int64)
        self.seg = torch.tensor([0, max(1, L // 4)], device=dev, dtype=torch.int64)
        self.L = L
        # warm-up runs real steps (on the zero-filled static inputs): put weights, optimizer state and BatchNorm
        # statistics back afterwards, so constructing a GraphedStep leaves the training state untouched
        keep = (model.flat.clone(), opt.m.clone(), opt.v.clone(), opt.steps, {k: v.clone() for k, v in model.buffers.items()})
        if warmup < 1:
            raise ValueError("GraphedStep needs at least one warm-up step: workspaces, library handles and the autograd "
                             "graph's accumulators must exist before the capture")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # library handles, workspaces and autotuning settle outside the capture
            for _ in range(warmup):
                self._fwd_bwd()
                opt.prepare()
                opt.step_captured()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.g1, self.g2 = torch.cuda.CUDAGraph(keep_graph=True), None
        # captured on the warm-up's stream: the leaves' gradient accumulators were created there (the estimator keeps
        # re-laid weights that hold them), and autograd warns -- and synchronises -- when a later backward feeds them
        # from another stream
        if self.split:
            with torch.cuda.graph(self.g1, stream=side):
                self._fwd_bwd()
            self.g2 = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(self.g2, pool=self.g1.pool(), stream=side):
                opt.step_captured()
        else:
            with torch.cuda.graph(self.g1, stream=side):
                self._fwd_bwd()
                opt.step_captured()
        # a captured memset replays with a corrupted fill value on this stack (count_memset_nodes): a torch reduction
        # that goes multi-block at this (B, L), or a library call that clears a flag buffer, would train on stale sums
        # from the second step on -- refuse the bucket instead
        self.nodes = 0
        for g in (self.g1, self.g2):
            if g is not None:
                memsets, total = count_memset_nodes(g)
                self.nodes += total
                if memsets:
                    raise RuntimeError(f"the captured step for B={B}, L={L} holds {memsets} memset node(s) of {total}: "
                                       "hipGraph memset nodes replay wrongly on this ROCm stack (see "
                                       "training.count_memset_nodes); run this bucket eagerly")
        model.flat.copy_(keep[0]), opt.m.copy_(keep[1]), opt.v.copy_(keep[2])
        opt.steps = keep[3]
        for k, v in keep[4].items():
            model.buffers[k].copy_(v)

    def _fwd_bwd(self):
        draws = {"seg": self.seg}
        if self.tz is not None:
            draws.update(t=self.tz[0], z=self.tz[1])
        ret = self.model(self.x, self.lens, self.logmel, self.midi, self.lft, draws=draws)
        self.cfm, self.prior = ret["cfm_loss"].detach(), ret["prior_loss"].detach()
        self.model.store.backward_into_flat(ret["cfm_loss"] + ret["prior_loss"])

    def __call__(self, x, lengths, logmel, midi, lft, segment=None):
        """returns (cfm_loss, prior_loss, grad_norm) as device tensors of the step just replayed"""
        for dst, src in ((self.x, x), (self.logmel, logmel), (self.midi, midi), (self.lft, lft), (self.lens, lengths)):
            dst.copy_(src, non_blocking=True)
        s0, n = self.model.draw_segment(self.L) if segment is None else segment
        self.seg.copy_(torch.tensor([s0, n], dtype=torch.int64))
        self.opt.prepare()
        self.g1.replay()
        if self.split:
            for a, b in self.sync.buckets:
                self.sync.dist.all_reduce(self.model.flat_grad[a:b], group=self.sync.group)
            if self.sync.world > 1:
                self.model.flat_grad.div_(self.sync.world)
            self.g2.replay()
        return self.cfm, self.prior, self.opt.norm
