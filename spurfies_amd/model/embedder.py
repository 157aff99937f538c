"""NeRF positional encoding with the reference's interface (spurfies/model/embedder.py:32-49):
get_embedder(multires, input_dims) -> (embed_fn, out_dim); output layout
[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]."""
import torch


class Embedder:
    def __init__(self, multires: int, input_dims: int = 3, include_input: bool = True):
        self.n_freqs = multires
        self.include_input = include_input
        self.freqs = [float(f) for f in (2.0 ** torch.linspace(0.0, multires - 1, multires)).tolist()]
        self.out_dim = input_dims * ((1 if include_input else 0) + 2 * multires)

    def embed(self, x):
        parts = [x] if self.include_input else []
        for f in self.freqs:
            xf = x * f
            parts.append(torch.sin(xf))
            parts.append(torch.cos(xf))
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    e = Embedder(multires, input_dims)
    return e.embed, e.out_dim
