"""Minimal stand-in for `torch_complex` (ComplexTensor container used by the reference frontend).
Test infrastructure only."""
from . import functional, tensor  # noqa: F401
