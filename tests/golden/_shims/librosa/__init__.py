"""Stand-in for `librosa` exposing only filters.mel (restated in oracle/mel.py from librosa's
published algorithm; librosa>=0.10.2 is an un-vendored dependency of the reference, pyproject.toml:40)."""
from . import filters  # noqa: F401
