"""Length masks — host-side mirror of serenade/utils/masking.py:4-210 (same signatures)."""
import torch


def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    """True on padded positions.  lengths: LongTensor (B,) or list.  See masking.py:4-121."""
    if length_dim == 0:
        raise ValueError("length_dim cannot be 0: {}".format(length_dim))
    if not isinstance(lengths, list):
        lengths = lengths.long().tolist()
    bs = len(lengths)
    if maxlen is None:
        maxlen = int(max(lengths)) if xs is None else xs.size(length_dim)
    else:
        assert xs is None
        assert maxlen >= int(max(lengths))
    pos = torch.arange(0, maxlen, dtype=torch.int64).unsqueeze(0).expand(bs, maxlen)
    mask = pos >= torch.tensor(lengths, dtype=torch.int64).unsqueeze(-1)
    if xs is not None:
        assert xs.size(0) == bs, (xs.size(0), bs)
        if length_dim < 0:
            length_dim = xs.dim() + length_dim
        ind = tuple(slice(None) if i in (0, length_dim) else None for i in range(xs.dim()))
        mask = mask[ind].expand_as(xs).to(xs.device)
    return mask


def make_non_pad_mask(lengths, xs=None, length_dim=-1):
    """True on valid positions (masking.py:124-210)."""
    return ~make_pad_mask(lengths, xs, length_dim)
