"""get_embedder(multires, i) -> (fn, out_dim): plain Fourier embedding with the raw input
(reference embedders/fourier.py:34-48).  Its output is never consumed by occnerf_mlp
(occnerf_mlp.py:180), so the renderer does not evaluate it; kept for the plug-in surface."""
import torch


def get_embedder(multires, i=0, input_dims=3):
    if i == -1:
        return torch.nn.Identity(), input_dims

    def embed(x):
        out = [x]
        for j in range(multires):
            f = float(2 ** j)
            out += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(out, -1)
    return embed, input_dims * (1 + 2 * multires)
