from serenade_amd import _shapes


def hifigan_state_shapes(**params):
    return _shapes.as_meta(_shapes.hifigan_shapes(**params))
