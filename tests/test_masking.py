"""serenade_amd.utils.masking (own body) against the reference's mask tables: tests/golden/masks.npz and the
xs / length_dim / maxlen forms captured from the reference by tests/golden/make_golden_post.py (masks_xs.npz)."""
import os

import numpy as np
import pytest
import torch

from serenade_amd.utils.masking import make_non_pad_mask, make_pad_mask

G = os.path.join(os.path.dirname(__file__), "golden")


def test_tables_from_the_reference_docstrings():
    g = np.load(os.path.join(G, "masks.npz"))
    lens = g["lengths"].tolist()
    for lengths in (lens, torch.tensor(lens), torch.tensor(lens, dtype=torch.int32)):
        assert np.array_equal(make_pad_mask(lengths).numpy(), g["pad"])
        assert np.array_equal(make_non_pad_mask(lengths).numpy(), g["non_pad"])
    assert make_pad_mask(lens).dtype == torch.bool


def test_xs_length_dim_and_maxlen_forms():
    g = np.load(os.path.join(G, "masks_xs.npz"))
    lens = g["lengths"].tolist()
    z = torch.zeros
    assert np.array_equal(make_pad_mask(lens, z(3, 4, 6)).numpy(), g["xs_3_4_6_dim-1"])
    assert np.array_equal(make_pad_mask(lens, z(3, 6, 6), 1).numpy(), g["xs_3_6_6_dim1"])
    assert np.array_equal(make_pad_mask(lens, z(3, 6, 6), 2).numpy(), g["xs_3_6_6_dim2"])
    assert np.array_equal(make_pad_mask(lens, z(3, 2, 5, 4), -2).numpy(), g["xs_3_2_5_4_dim-2"])
    assert np.array_equal(make_pad_mask(lens, maxlen=7).numpy(), g["maxlen7"])
    assert np.array_equal(make_pad_mask(torch.tensor(lens)).numpy(), g["tensor_lengths"])
    assert np.array_equal(make_non_pad_mask(lens, z(3, 4, 6)).numpy(), g["non_pad_xs_3_4_6"])
    assert np.array_equal(make_non_pad_mask(lens, z(3, 6, 6), 1).numpy(), g["non_pad_xs_3_6_6_dim1"])


def test_argument_errors():
    with pytest.raises(ValueError):
        make_pad_mask([3, 2], torch.zeros(2, 4), length_dim=0)
    with pytest.raises(AssertionError):
        make_pad_mask([3, 2], maxlen=2)
    with pytest.raises(AssertionError):
        make_pad_mask([3, 2], torch.zeros(2, 4), maxlen=4)
    with pytest.raises(AssertionError):
        make_pad_mask([3, 2], torch.zeros(3, 4))
