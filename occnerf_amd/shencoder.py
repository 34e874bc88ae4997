"""Surface of the reference's spherical-harmonics encoder module (core/nets/occnerf/shencoder/sphere_harmonics.py:62-87).

The rendering path never evaluates it (occnerf_mlp.py:46 is commented out in the reference); the class keeps the constructor
and attributes so code that constructs one still runs, and its forward refuses by name through `_shencoder`."""
import torch.nn as nn


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, 'SH encoder only support input dim == 3'
        assert 0 < self.degree <= 8, 'SH encoder only supports degree in [1, 8]'

    def __repr__(self):
        return f'SHEncoder: input_dim={self.input_dim} degree={self.degree}'

    def forward(self, inputs, size=1):
        import _shencoder
        _shencoder.sh_encode_forward(inputs, None, inputs.shape[0], self.input_dim, self.degree, None)
