"""Two-column text tables (`wav.scp`, key files): espnet2/fileio/read_text.py:8-35."""
import logging
from pathlib import Path
from typing import Dict, Union


def read_2columns_text(path: Union[Path, str]) -> Dict[str, str]:
    """`key value with spaces` per line -> {key: value}; a line without a value maps to ""; a
    repeated key is an error (read_text.py:24-33)."""
    data: Dict[str, str] = {}
    with Path(path).open("r", encoding="utf-8") as f:
        for linenum, line in enumerate(f, 1):
            sps = line.rstrip().split(maxsplit=1)
            if not sps:
                logging.warning(f"empty line at {path}:L{linenum}")
                continue
            k, v = (sps[0], "") if len(sps) == 1 else sps
            if k in data:
                raise RuntimeError(f"{k} is duplicated ({path}:{linenum})")
            data[k] = v
    return data
