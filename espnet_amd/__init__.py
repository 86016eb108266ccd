"""espnet_amd — MI355X-native (gfx950) implementation of ESPnet's ASR-inference hot path:
STFT/log-mel frontend -> Conformer encoder -> CTC / joint CTC-attention decoding, behind the
`espnet2.bin.asr_inference.Speech2Text` / `espnet2.tasks.asr.ASRTask` plugin API.

Python here is host plumbing only (lengths, packing weights once, launching); every number on the
hot path is produced by the hand-written HIP kernels in espnet_amd/csrc through the C ABI of
include/espnet_amd.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
