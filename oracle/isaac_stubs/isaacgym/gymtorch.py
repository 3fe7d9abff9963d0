"""Empty name-holder: the golden generator never wraps simulator buffers."""


def unwrap_tensor(t):
    """Isaac Gym hands the sim a raw device pointer; the generator's dummy sim takes the tensor."""
    return t


def wrap_tensor(t):
    return t
