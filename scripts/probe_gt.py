"""dev probe: brute-force ground truth agreement (engine vs torch full-row topk vs torch chunked topk)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cosdata_amd as ca
n = int(sys.argv[1]); d = int(sys.argv[2]); B = 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
nc = max(64, n // 1000)
centers = torch.randn(nc, d, generator=g, device=dev); centers /= centers.norm(dim=1, keepdim=True)
X = torch.empty(n, d, device=dev)
for s in range(0, n, 1 << 18):
    m = min(1 << 18, n - s)
    x = centers[torch.randint(0, nc, (m,), generator=g, device=dev)] + (0.8 / d ** 0.5) * torch.randn(m, d, generator=g, device=dev)
    X[s:s + m] = x / x.norm(dim=1, keepdim=True)
x = centers[torch.randint(0, nc, (B,), generator=g, device=dev)] + (0.8 / d ** 0.5) * torch.randn(B, d, generator=g, device=dev)
Q = x / x.norm(dim=1, keepdim=True)
ix = ca.HNSWIndex(d); ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
mine, msc = ix.bruteforce_topk(Q.cpu().numpy(), 10)
full = (Q @ X.T).topk(10, dim=1)
ci, cs = [], []
for s in range(0, n, 1 << 20):
    t = (Q @ X[s:s + (1 << 20)].T).topk(10, dim=1)
    ci.append(t.indices + s); cs.append(t.values)
ci, cs = torch.cat(ci, 1), torch.cat(cs, 1)
sel = cs.topk(10, dim=1).indices
chunked = torch.gather(ci, 1, sel).cpu().numpy()
fulli = full.indices.cpu().numpy()
agree = lambda a, b: np.mean([len(set(a[i].tolist()) & set(b[i].tolist())) / 10 for i in range(B)])
print(f"n={n} d={d}: mine~full {agree(mine, fulli):.4f}  mine~chunked {agree(mine, chunked):.4f}  full~chunked {agree(fulli, chunked):.4f}")
print("mine[0]", mine[0], msc[0][:3]); print("full[0]", fulli[0], full.values[0][:3].cpu().numpy()); print("chunk[0]", chunked[0])
