"""Laplace-CDF density with the reference's interface (spurfies/model/density.py:5-30)."""
import torch
import torch.nn as nn


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for name, value in dict(params_init).items():
            setattr(self, name, nn.Parameter(torch.tensor(float(value))))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    """sigma = (1/beta) * (0.5 + 0.5 * sign(s) * expm1(-|s|/beta)),  beta = |beta_param| + beta_min."""

    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.register_buffer("beta_min", torch.tensor(float(beta_min)), persistent=False)
        self.beta_min_value = float(beta_min)            # host copy: reading the buffer back would synchronise (and cannot be captured)

    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def get_beta_value(self):
        """get_beta().detach(), computed once per parameter version (the sampler and the compositing kernel of one step read the
        same value: two launches instead of four)."""
        ext = getattr(self, "_beta_forward", None)
        if ext is not None:              # the model's forward formed it in its first launch (ops.camera_rays) for this forward
            return ext
        key = (self.beta._version, self.beta.data_ptr())
        capturing = self.beta.is_cuda and torch.cuda.is_current_stream_capturing()     # a graph replay must recompute it itself
        if capturing or getattr(self, "_beta_value_key", None) != key:
            with torch.no_grad():
                self._beta_value = self.beta.abs() + self.beta_min
            self._beta_value_key = None if capturing else key
        return self._beta_value
