"""Placeholder: nothing on the ASR hot path calls into torch_complex.functional."""
