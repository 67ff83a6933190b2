import torch
dev = torch.device("cuda:0"); n = 8192
a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
for _ in range(3): c = a.t() @ b
torch.cuda.synchronize()
