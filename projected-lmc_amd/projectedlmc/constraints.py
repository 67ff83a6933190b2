"""Parameter constraints with gpytorch's semantics [gpytorch-knowledge]: value =
softplus(raw) (+ lower bound).  Used by kernels (Positive) and likelihoods (GreaterThan),
e.g. `gp.constraints.GreaterThan(np.exp(noise_thresh))` at projected_lmc.py:921."""
import torch
import torch.nn.functional as F


class Positive(torch.nn.Module):
    lower_bound = 0.0

    def transform(self, raw):
        return F.softplus(raw) + self.lower_bound

    def inverse_transform(self, value):
        v = torch.as_tensor(value) - self.lower_bound
        return v + torch.log(-torch.expm1(-v))


class GreaterThan(Positive):
    def __init__(self, lower_bound):
        super().__init__()
        self.lower_bound = float(lower_bound)

    def __repr__(self):
        return "GreaterThan(%.3E)" % self.lower_bound
