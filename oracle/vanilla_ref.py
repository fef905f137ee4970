"""TEST / BASELINE INFRASTRUCTURE -- restatement of the reference's pure-PyTorch CPU encoding + MLP path
(``models/network_utils.py:14-37`` ``VanillaFrequency``, ``:95-139`` ``VanillaMLP``, ``:68-79`` ``CompositeEncoding``):
BASELINE.json configs[0] ("VanillaMLP + pure-PyTorch encoding on CPU, no tcnn / nerfacc").  Used by ``bench.py``'s
``cpu_baseline`` leg (timed on the GPU box's host cores) and pinned against the reference modules themselves by
tests/test_oracle_vanilla.py (build container) and the committed fixture tests/golden/vanilla_frequency.npz.
Never imported by the product package."""
import math

import torch
import torch.nn as nn


class VanillaFrequency(nn.Module):
    """out = cat_k [sin(2^k x) m_k, cos(2^k x) m_k]  (network_utils.py:14-37; the cosine coarse-to-fine mask m_k of
    ``n_masking_step`` is all ones when masking is off)"""

    def __init__(self, in_channels, config):
        super().__init__()
        self.N_freqs = config["n_frequencies"]
        self.n_input_dims = in_channels
        self.freq_bands = 2 ** torch.linspace(0, self.N_freqs - 1, self.N_freqs)
        self.n_output_dims = in_channels * 2 * self.N_freqs
        self.n_masking_step = config.get("n_masking_step", 0)
        self.update_step(None, None)

    def forward(self, x):
        out = []
        for freq, mask in zip(self.freq_bands, self.mask):
            out += [torch.sin(freq * x) * mask, torch.cos(freq * x) * mask]
        return torch.cat(out, -1)

    def update_step(self, epoch, global_step):
        if self.n_masking_step <= 0 or global_step is None:
            self.mask = torch.ones(self.N_freqs, dtype=torch.float32)
        else:
            ramp = (global_step / self.n_masking_step * self.N_freqs - torch.arange(0, self.N_freqs)).clamp(0, 1)
            self.mask = (1.0 - torch.cos(math.pi * ramp)) / 2.0


class VanillaMLP(nn.Module):
    """Linear(+bias) stack, ReLU (kaiming-uniform) -- the non-sphere-init branch of network_utils.py:95-139"""

    def __init__(self, dim_in, dim_out, n_neurons=64, n_hidden_layers=1):
        super().__init__()
        dims = [dim_in] + [n_neurons] * n_hidden_layers + [dim_out]
        layers = []
        for i in range(len(dims) - 1):
            lin = nn.Linear(dims[i], dims[i + 1], bias=True)
            nn.init.constant_(lin.bias, 0.0)
            nn.init.kaiming_uniform_(lin.weight, nonlinearity="relu")
            layers.append(lin)
            if i < len(dims) - 2:
                layers.append(nn.ReLU(inplace=True))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        return self.layers(x.float())


def include_xyz(encoding, x):
    """CompositeEncoding(include_xyz=True, xyz_scale=2, xyz_offset=-1), network_utils.py:75-76"""
    return torch.cat([x * 2.0 - 1.0, encoding(x)], dim=-1)
