from serenade_amd import _shapes


def serenade_state_shapes(**params):
    return _shapes.as_meta(_shapes.serenade_shapes(**params))
