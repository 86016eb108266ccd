"""UtteranceMVN (espnet2/layers/utterance_mvn.py:10-88), task default norm_means=True,
norm_vars=False.  On the MI355X path the mean subtraction is fused into the first subsampling
conv (csrc/frontend.hip): this module only records the configuration and, when called on its own,
produces the per-utterance partial sums the fused kernel consumes."""
import torch

from espnet_amd import lib as L


class UtteranceMVN(torch.nn.Module):
    def __init__(self, norm_means: bool = True, norm_vars: bool = False, eps: float = 1.0e-20):
        super().__init__()
        if not norm_means or norm_vars:
            raise NotImplementedError(
                "UtteranceMVN(norm_means=True, norm_vars=False) is the only fused configuration")
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps

    def extra_repr(self):
        return f"norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def partial_sums(self, feats: torch.Tensor, flens_dev: torch.Tensor) -> torch.Tensor:
        """feats (B, T_f, D) f32 on the GPU, flens_dev (B,) i32 on the GPU -> (B, 8, D) f32."""
        L.require_gpu(feats, "feats")
        B, T_f, D = feats.shape
        partial = torch.empty(B, 8, D, dtype=torch.float32, device=feats.device)
        L.check(L.load().em_utt_mvn_partial_f32(L.ptr(feats), L.ptr(flens_dev), B, T_f, D,
                                                L.ptr(partial), L.current_stream_ptr()),
                "em_utt_mvn_partial_f32")
        return partial

    def forward_device(self, feats: torch.Tensor, flens_dev: torch.Tensor) -> torch.Tensor:
        """Stand-alone normalisation (the streaming frontend needs the features themselves):
        feats (B, T_f, D) f32 on the GPU, IN PLACE like the reference (`x -= mean`, :73)."""
        partial = self.partial_sums(feats, flens_dev)
        B, T_f, D = feats.shape
        L.check(L.load().em_utt_mvn_apply_f32(L.ptr(feats), L.ptr(partial), L.ptr(flens_dev), B, T_f, D,
                                              L.current_stream_ptr()), "em_utt_mvn_apply_f32")
        return feats

    def forward(self, x: torch.Tensor, ilens: torch.Tensor = None):
        if ilens is None:
            ilens = x.new_full([x.size(0)], x.size(1))
        flens_dev = ilens.to(device=x.device, dtype=torch.int32)
        return self.forward_device(x.to(torch.float32).contiguous(), flens_dev), ilens
