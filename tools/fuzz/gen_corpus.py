import sys, numpy as np, struct
import os; REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO)
from tests.test_cpu_flac import _speechlike, _plans
from tests.test_cpu_cli import _riff
from tests.flac_writer import encode_flac
x = _speechlike(3000, 1)
plans = dict((n, (bs, p)) for n, bs, p in _plans(x, 16))
blobs = [encode_flac([x], 16, 16000, *plans[k]) for k in ("mixed_300", "lpc12_p15", "fixed2_escape", "lpc8_rice2", "tiny_blocks_many_frames", "lpc32_p14")]
x24 = _speechlike(2000, 2, 24)
blobs.append(encode_flac([x24], 24, 16000, 512, lambda fi, ci, s, b: dict(kind="fixed", order=3, po=2, method=1)))
rng = np.random.default_rng(7)
blobs.append(_riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 3000).astype("<i2").tobytes(), junk=b"LIST" + struct.pack("<I", 3) + b"abc\x00"))
blobs.append(_riff(1, 1, 16000, 24, bytes(rng.integers(0, 256, 3000).astype(np.uint8)), extensible=True))
blobs.append(_riff(3, 1, 16000, 64, rng.normal(0, 1, 500).astype("<f8").tobytes()))
n = 0
for it in range(4000):
    b = bytearray(blobs[it % len(blobs)])
    for _ in range(int(rng.integers(0, 6))):
        pos = int(rng.integers(0, len(b)))
        if rng.random() < 0.5: b[pos] = int(rng.integers(0, 256))
        else: b[pos] ^= 1 << int(rng.integers(0, 8))
    if rng.random() < 0.15: b = b[: int(rng.integers(0, len(b)))]
    if rng.random() < 0.05:
        a, c = sorted(int(v) for v in rng.integers(0, len(b) + 1, 2)); del b[a:c]
    open(f"/tmp/fz/{n:05d}.bin", "wb").write(bytes(b)); n += 1
print(n)
