import numpy as np, torch
rng = np.random.default_rng(0)
x = rng.uniform(0.01, 50.0, 4_000_000)
d = torch.from_numpy(x).cuda()
print("sqrt mismatches:", int((torch.sqrt(d).cpu().numpy() != np.sqrt(x)).sum()), "of", len(x))
y = rng.uniform(0.01, 50.0, 4_000_000)
print("div mismatches:", int(((d / torch.from_numpy(y).cuda()).cpu().numpy() != x / y).sum()))
xi = np.arange(1, 2_000_000, dtype=np.float64)
print("sqrt(int) mismatches:", int((torch.sqrt(torch.from_numpy(xi).cuda()).cpu().numpy() != np.sqrt(xi)).sum()))
