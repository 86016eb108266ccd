"""Kaldi-style result directory writer — the on-disk contract of `asr.sh` stage 12
(espnet2/fileio/datadir_writer.py:8-75): `writer["1best_recog"]["token"]["utt1"] = "a b"` appends
the line `utt1 a b` to `<dir>/1best_recog/token`.  A node is either a directory (has children) or a
file (has lines), never both; duplicate keys and mismatching key sets between sibling files warn."""
import warnings
from pathlib import Path
from typing import Dict, Optional, Set, TextIO, Union


class DatadirWriter:
    def __init__(self, p: Union[Path, str]):
        self.path = Path(p)
        self.children: Dict[str, "DatadirWriter"] = {}
        self.fd: Optional[TextIO] = None
        self.keys: Set[str] = set()

    @property
    def has_children(self) -> bool:
        return bool(self.children)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def __getitem__(self, key: str) -> "DatadirWriter":
        if not isinstance(key, str):
            raise TypeError(f"key must be str, got {type(key)}")
        if self.fd is not None:
            raise RuntimeError("This writer points out a file")
        node = self.children.get(key)
        if node is None:
            node = self.children[key] = DatadirWriter(self.path / key)
        return node

    def __setitem__(self, key: str, value: str):
        if not isinstance(key, str) or not isinstance(value, str):
            raise TypeError("key and value must be str")
        if self.children:
            raise RuntimeError("This writer points out a directory")
        if key in self.keys:
            warnings.warn(f"Duplicated: {key}")
        if self.fd is None:
            self.path.parent.mkdir(parents=True, exist_ok=True)
            self.fd = self.path.open("w", encoding="utf-8")
        self.keys.add(key)
        self.fd.write(f"{key} {value}\n")

    def flush(self):
        for c in self.children.values():
            c.flush()
        if self.fd is not None:
            self.fd.flush()

    def close(self):
        prev = None
        for child in self.children.values():
            child.close()
            if prev is not None and prev.keys != child.keys:
                warnings.warn(f"Ids are mismatching between {prev.path} and {child.path}")
            prev = child
        if self.fd is not None:
            self.fd.close()
            self.fd = None
