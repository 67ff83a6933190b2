"""Mean functions: ZeroMean / ConstantMean / MultitaskMean (what the BASELINE configs use) and the reference's own
LinearMean / PolynomialMean (projected_lmc.py:38-81), with the
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


class LinearMean(Mean):
    """m(x) = x w + b with weights (*batch, d, 1) and bias (*batch, 1), randn-initialised (projected_lmc.py:65-81)."""

    def __init__(self, input_size, batch_shape=torch.Size(), bias=True, **kwargs):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)
        self.register_parameter("weights", torch.nn.Parameter(torch.randn(*self.batch_shape, input_size, 1)))
        if bias:
            self.register_parameter("bias", torch.nn.Parameter(torch.randn(*self.batch_shape, 1)))
        else:
            self.bias = None

    def forward(self, x):
        res = x.matmul(self.weights.to(x.dtype)).squeeze(-1)
        if self.bias is not None:
            res = res + self.bias.to(x.dtype)
        return res

    def basis_matrix(self, x):
        return torch.hstack([x, torch.ones((len(x), 1), device=x.device, dtype=x.dtype)])


class PolynomialMean(Mean):
    """m(x) = sum_{i=1..degree} (x ** i) w_i + b (projected_lmc.py:38-63; like the reference, a `weights_0` parameter is
    registered and never used)."""

    def __init__(self, input_size, batch_shape=torch.Size(), bias=True, degree=3, **kwargs):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)
        for i in range(degree + 1):
            self.register_parameter("weights_{0}".format(i),
                                    torch.nn.Parameter(torch.randn(*self.batch_shape, input_size, 1)))
        if bias:
            self.register_parameter("bias", torch.nn.Parameter(torch.randn(*self.batch_shape, 1)))
        else:
            self.bias = None
        self.degree = degree

    def forward(self, x):
        res = 0
        for i in range(1, self.degree + 1):
            res = res + (x ** i).matmul(getattr(self, "weights_{0}".format(i)).to(x.dtype)).squeeze(-1)
        if self.bias is not None:
            res = res + self.bias.to(x.dtype)
        return res


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
