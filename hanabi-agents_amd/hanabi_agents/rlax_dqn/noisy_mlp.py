"""Q-network: NoisyLinear / NoisyMLP in PyTorch (the GEMMs run on MFMA through torch.matmul).

Arithmetic of the reference layer (hanabi_agents/rlax_dqn/noisy_mlp.py:61-91):

    y = x @ w + b  +  x @ (w_mu + w_sigma * eps_w)  +  (b_mu + b_sigma * eps_b)

with six parameters per layer (w, b, w_mu, b_mu, w_sigma, b_sigma), all three
weight matrices initialised TruncatedNormal(stddev = 1/sqrt(fan_in)) and all biases zero
(noisy_mlp.py:43-47,57-78), ReLU between layers and none after the last (noisy_mlp.py:176-185).

MI355X form: the two products share x, so they are ONE GEMM against the effective weight
W = w + w_mu + w_sigma * eps_w (an elementwise pass over [in, out], 0.86 M elements for the
2-player net) — half the matrix FLOPs of the literal form; autograd distributes the gradient
back to w, w_mu and w_sigma. `merged=False` evaluates the literal two-GEMM form (parity tests).

Noise: in the reference the noise keys are fixed at trace time, so eps_w and
eps_b are the SAME tensors on every call; that is the default here (`frozen` buffers drawn once
from the layer's own generator). `resample()` draws fresh noise; explicit tensors can be passed
to `forward` for parity tests. eps_b may be [out] or [batch, out] (the reference draws it after
broadcasting, noisy_mlp.py:83-87).
"""
import math
from typing import Iterable, Optional, Sequence

import torch
from torch import nn


def _trunc_normal_(t: torch.Tensor, std: float, gen: torch.Generator):
    # hk.initializers.TruncatedNormal(stddev): stddev * truncated_normal(-2, 2)
    tmp = torch.empty(t.shape, dtype=torch.float32)
    nn.init.trunc_normal_(tmp, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    with torch.no_grad():
        t.copy_(tmp * std)
    return t


class NoisyLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, seed: int = 0, with_bias: bool = True,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.in_features, self.out_features, self.with_bias = in_features, out_features, with_bias
        self.compute_dtype = compute_dtype
        gen = torch.Generator().manual_seed(seed)
        std = 1.0 / math.sqrt(in_features)
        self.w = nn.Parameter(_trunc_normal_(torch.empty(in_features, out_features), std, gen))
        self.w_mu = nn.Parameter(_trunc_normal_(torch.empty(in_features, out_features), std, gen))
        self.w_sigma = nn.Parameter(_trunc_normal_(torch.empty(in_features, out_features), std, gen))
        if with_bias:
            self.b = nn.Parameter(torch.zeros(out_features))
            self.b_mu = nn.Parameter(torch.zeros(out_features))
            self.b_sigma = nn.Parameter(torch.zeros(out_features))
        self._noise_gen = torch.Generator().manual_seed(seed + 7919)
        self.register_buffer("eps_w", torch.randn(in_features, out_features, generator=self._noise_gen))
        self.register_buffer("eps_b", torch.randn(out_features, generator=self._noise_gen))

    def resample(self):
        """Draw fresh independent Gaussian noise (per-step resampling mode)."""
        self.eps_w.normal_()  # on the buffers' own device: no host round trip
        self.eps_b.normal_()

    def effective(self, eps_w=None, eps_b=None):
        """(W, bias) of the merged single-GEMM form."""
        eps_w = self.eps_w if eps_w is None else eps_w
        W = self.w + self.w_mu + self.w_sigma * eps_w
        if not self.with_bias:
            return W, None
        eps_b = self.eps_b if eps_b is None else eps_b
        return W, self.b + self.b_mu + self.b_sigma * eps_b

    def forward(self, x, eps_w=None, eps_b=None, merged: bool = True):
        cd = self.compute_dtype
        if merged:
            W, bias = self.effective(eps_w, eps_b)
            y = torch.matmul(x.to(cd), W.to(cd)).to(W.dtype)
            return y if bias is None else y + bias
        eps_w = self.eps_w if eps_w is None else eps_w
        y = torch.matmul(x.to(cd), self.w.to(cd)).to(self.w.dtype)
        y_noisy = torch.matmul(x.to(cd), (self.w_mu + self.w_sigma * eps_w).to(cd)).to(self.w.dtype)
        if self.with_bias:
            eps_b = self.eps_b if eps_b is None else eps_b
            y = y + self.b
            y_noisy = y_noisy + (self.b_mu + self.b_sigma * eps_b)
        return y + y_noisy


class NoisyMLP(nn.Module):
    """`NoisyMLP(output_sizes)` of the reference: NoisyLinear layers with ReLU in between."""

    def __init__(self, input_size: int, output_sizes: Iterable[int], seed: int = 1234, with_bias: bool = True,
                 activate_final: bool = False, compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        sizes = [input_size] + list(output_sizes)
        self.layers = nn.ModuleList(
            NoisyLinear(sizes[i], sizes[i + 1], seed=seed + 101 * i, with_bias=with_bias, compute_dtype=compute_dtype)
            for i in range(len(sizes) - 1))
        self.activate_final = activate_final

    def resample(self):
        for layer in self.layers:
            layer.resample()

    def forward(self, x, noise: Optional[Sequence] = None, merged: bool = True):
        """noise: optional list of (eps_w, eps_b) per layer."""
        out = x
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            ew, eb = (None, None) if noise is None else noise[i]
            out = layer(out, ew, eb, merged=merged)
            if i < last or self.activate_final:
                out = torch.relu(out)
        return out

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())


class PlainMLP(nn.Module):
    """hk.nets.MLP of the older scalar-Q agent (hanabi_agents/rlax_dqn/rlax_dqn.py:26-33), BASELINE config 2."""

    def __init__(self, input_size: int, output_sizes: Iterable[int], seed: int = 1234,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        sizes = [input_size] + list(output_sizes)
        gen = torch.Generator().manual_seed(seed)
        self.compute_dtype = compute_dtype
        self.weights = nn.ParameterList()
        self.biases = nn.ParameterList()
        for i in range(len(sizes) - 1):
            self.weights.append(nn.Parameter(_trunc_normal_(torch.empty(sizes[i], sizes[i + 1]), 1.0 / math.sqrt(sizes[i]), gen)))
            self.biases.append(nn.Parameter(torch.zeros(sizes[i + 1])))

    def resample(self):
        pass

    def forward(self, x, noise=None, merged=True):
        out = x
        last = len(self.weights) - 1
        for i, (w, b) in enumerate(zip(self.weights, self.biases)):
            out = torch.matmul(out.to(self.compute_dtype), w.to(self.compute_dtype)).to(w.dtype) + b
            if i < last:
                out = torch.relu(out)
        return out

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())
