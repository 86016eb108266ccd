"""ORACLE / test infrastructure: deterministic "recipe" weights shared by the golden generator
(tests/golden/make_golden.py, which loads them INTO the reference classes), the oracle and the
HIP path, so fixtures need not carry tens of MB of parameters.

Every tensor is drawn from its own CPU `torch.Generator` seeded by (seed, crc32(key)), so the
result depends only on (key, shape, seed) and on the pinned torch version of this image.
Scales follow torch's default initialisers (uniform +-1/sqrt(fan_in) for Linear/Conv, N(0,1)
for Embedding, xavier for pos_bias_{u,v}); LayerNorm/BatchNorm affine terms and BatchNorm
running statistics are perturbed away from (1, 0, 0, 1) so that parity tests exercise them.
"""
import zlib

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2**63 - 1))
    return g


def _uniform(shape, bound, g):
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0) * bound


def recipe_tensor(key: str, shape, seed: int) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _gen(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 1)[0] if "." in key else ""
    pleaf = parent.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    is_norm = pleaf.startswith("norm") or pleaf == "after_norm"
    if is_norm and leaf == "weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if is_norm and leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf in ("pos_bias_u", "pos_bias_v"):
        bound = (6.0 / (shape[0] + shape[1])) ** 0.5
        return _uniform(shape, bound, g)
    if key.endswith("embed.0.weight") and len(shape) == 2 and "decoder" in key:
        return torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return _uniform(shape, 1.0 / fan_in**0.5, g)
    if leaf == "bias":
        return _uniform(shape, 0.05, g)
    return torch.randn(shape, generator=g) * 0.1


def recipe_state_dict(shapes, seed: int, skip=("frontend.logmel.melmat",)):
    """shapes: mapping key -> shape.  Keys in `skip` are left to the caller (the mel matrix is
    computed, not random)."""
    out = {}
    for key in sorted(shapes):
        if key in skip:
            continue
        out[key] = recipe_tensor(key, shapes[key], seed)
    return out


def synth_waveform(utt_idx: int, n_samples: int = 160000) -> torch.Tensor:
    """BASELINE.md §3 input definition: N(0, 0.1^2) fp32, generator seed 1000 + utt_idx."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + int(utt_idx))
    return torch.randn(n_samples, generator=g, dtype=torch.float32) * 0.1


def token_list(vocab: int = 5000):
    return ["<blank>", "<unk>"] + [f"t{i}" for i in range(vocab - 3)] + ["<sos/eos>"]
