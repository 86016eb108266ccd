"""No-op stand-in for `typeguard` so the read-only reference imports in this container.

Test infrastructure only (used by tests/golden/make_golden.py); carries no arithmetic.
"""


def typechecked(func=None, **_kwargs):
    if func is None:
        return lambda f: f
    return func


def check_argument_types(*_a, **_k):
    return True


def check_return_type(*_a, **_k):
    return True
