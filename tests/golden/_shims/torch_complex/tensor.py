import torch


class ComplexTensor:
    """(real, imag) pair with the handful of methods espnet2/asr/frontend/default.py touches."""

    def __init__(self, real, imag=None):
        if imag is None:
            imag = torch.zeros_like(real)
        self.real = real
        self.imag = imag

    def dim(self):
        return self.real.dim()

    def size(self, *a):
        return self.real.size(*a)

    @property
    def shape(self):
        return self.real.shape

    @property
    def dtype(self):
        return self.real.dtype

    @property
    def device(self):
        return self.real.device

    def __len__(self):
        return len(self.real)

    def __getitem__(self, idx):
        return ComplexTensor(self.real[idx], self.imag[idx])
