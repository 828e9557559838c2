"""Length masks on the host.  Interface of serenade/utils/masking.py:4,124 (``make_pad_mask`` / ``make_non_pad_mask``:
same arguments, same results on the tables in its docstrings — tests/golden/masks.npz); the body is this repo's own.

A mask is `frame index >= length` along one axis of the result; with a reference tensor ``xs`` the comparison is laid
along ``xs``'s ``length_dim`` and broadcast over every other non-batch axis."""
import torch


def _lengths_column(lengths):
    if torch.is_tensor(lengths):
        lengths = lengths.detach().to("cpu", torch.int64)
    else:
        lengths = torch.as_tensor([int(v) for v in lengths], dtype=torch.int64)
    if lengths.dim() != 1 or lengths.numel() == 0:
        raise ValueError(f"lengths must be a non-empty 1-D sequence, got shape {tuple(lengths.shape)}")
    return lengths


def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    """bool tensor, True where a frame lies at or beyond its item's length.

    lengths: (B,) LongTensor or list.  Without ``xs`` the result is (B, maxlen or max(lengths)).  With ``xs`` the
    result has ``xs``'s shape and device and the frame axis is ``length_dim`` (which may not be the batch axis 0)."""
    lens = _lengths_column(lengths)
    n_items = lens.numel()
    longest = int(lens.max())
    if xs is None:
        width = longest if maxlen is None else int(maxlen)
        if width < longest:
            raise AssertionError(f"maxlen {width} is shorter than the longest item ({longest})")
        return torch.arange(width).view(1, width) >= lens.view(n_items, 1)

    if maxlen is not None:
        raise AssertionError("maxlen and xs are mutually exclusive")
    axis = length_dim % xs.dim() if length_dim != 0 else 0
    if axis == 0:
        raise ValueError(f"length_dim cannot be 0: {length_dim}")
    if xs.size(0) != n_items:
        raise AssertionError(f"{n_items} lengths for a batch of {xs.size(0)}")
    width = xs.size(axis)
    shape = [1] * xs.dim()
    shape[axis] = width
    frame = torch.arange(width).view(shape)
    shape[axis], shape[0] = 1, n_items
    return (frame >= lens.view(shape)).expand(xs.shape).to(xs.device)


def make_non_pad_mask(lengths, xs=None, length_dim=-1):
    """bool tensor, True on the valid frames: the complement of ``make_pad_mask``."""
    return make_pad_mask(lengths, xs, length_dim).logical_not()
