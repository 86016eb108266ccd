"""Import-only stub so the reference module graph loads; never called on the ASR path."""


def jit(*a, **k):  # noqa: E302
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


njit = jit
