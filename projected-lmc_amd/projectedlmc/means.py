"""Mean functions the BASELINE configs use (ZeroMean / ConstantMean / MultitaskMean), with the
constructor surface the reference relies on: `mean_type(input_size=dim, batch_shape=...)`
(projected_lmc.py:298,739,752) and `MultitaskMean(base, num_tasks=p)` (:460).
[gpytorch-knowledge] ConstantMean holds `raw_constant` (batch_shape), initialised to 0."""
import copy

import torch


class Mean(torch.nn.Module):
    pass


class ZeroMean(Mean):
    def __init__(self, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)

    def forward(self, x):
        return torch.zeros(*self.batch_shape, x.shape[-2], dtype=x.dtype, device=x.device)


class ConstantMean(Mean):
    def __init__(self, constant_prior=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)
        self.register_parameter("raw_constant", torch.nn.Parameter(torch.zeros(self.batch_shape)))

    @property
    def constant(self):
        return self.raw_constant

    def forward(self, x):
        c = self.raw_constant.to(x.dtype)
        return c.unsqueeze(-1).expand(*self.batch_shape, x.shape[-2])


class MultitaskMean(Mean):
    """p copies of a base mean -> (n, p)."""

    def __init__(self, base_means, num_tasks):
        super().__init__()
        if isinstance(base_means, Mean):
            base_means = [base_means] + [copy.deepcopy(base_means) for _ in range(num_tasks - 1)]
        if len(base_means) != num_tasks:
            raise RuntimeError("base_means should be a list of means of length num_tasks")
        self.base_means = torch.nn.ModuleList(base_means)
        self.num_tasks = num_tasks

    def forward(self, x):
        cols = []
        for m in self.base_means:
            v = m(x)
            cols.append(v.reshape(-1, x.shape[-2])[0] if v.dim() > 1 else v)
        return torch.stack(cols, dim=-1)
