"""Development probe (not part of the product): build time + search throughput at a given size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cosdata_amd as ca
from tests import helpers as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
t = time.time()
g = torch.Generator(device="cuda"); g.manual_seed(42)
nc = 1024
centers = torch.randn(nc, d, generator=g, device="cuda"); centers /= centers.norm(dim=1, keepdim=True)
assign = torch.randint(0, nc, (n,), generator=g, device="cuda")
X = centers[assign] + 0.1 * torch.randn(n, d, generator=g, device="cuda") / (d ** 0.5) * 8
X /= X.norm(dim=1, keepdim=True)
X = X.contiguous()
torch.cuda.synchronize()
print("gen", time.time() - t)
ix = ca.HNSWIndex(d)
t = time.time()
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
print("upload+quantize", time.time() - t)
t = time.time()
ix.build(bs)
print("build", time.time() - t, [ix.level_count(l) for l in range(10)])
B = 256
qi = torch.randint(0, n, (B * 16,), generator=g, device="cuda")
Q = (X[qi] + 0.02 * torch.randn(B * 16, d, generator=g, device="cuda") / (d ** 0.5) * 8).contiguous()
out_ids = torch.zeros(B * 16, 10, dtype=torch.int32, device="cuda")
out_sc = torch.zeros(B * 16, 10, dtype=torch.float32, device="cuda")
out_cnt = torch.zeros(B * 16, dtype=torch.int32, device="cuda")
out_st = torch.zeros(B * 16, dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(16)]
def run(nb, conc):
    torch.cuda.synchronize()
    t = time.time()
    for i in range(nb):
        s = streams[i % conc]; j = i % 16
        ix.batch_search_device(Q[j*B:(j+1)*B].data_ptr(), B, 10, out_ids[j*B:(j+1)*B].data_ptr(), out_sc[j*B:(j+1)*B].data_ptr(),
                               out_cnt[j*B:(j+1)*B].data_ptr(), out_st[j*B:(j+1)*B].data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return nb * B / (time.time() - t)
for ef in (64, 256):
    ix.set_ef_search(ef)
    run(2, 1)
    for conc in (1, 4, 16):
        print(f"ef={ef} conc={conc} qps={run(32, conc):.0f}")
    # recall vs torch brute force on 512 queries
    sims = Q[:512] @ X.T
    gt = sims.topk(10, dim=1).indices.cpu().numpy()
    ids = out_ids[:512].cpu().numpy()
    # out_ids hold results of the last batches for those slots
    ix.batch_search_device(Q[:256].data_ptr(), 256, 10, out_ids[:256].data_ptr(), out_sc[:256].data_ptr(), out_cnt[:256].data_ptr(), out_st[:256].data_ptr(), 0)
    ix.batch_search_device(Q[256:512].data_ptr(), 256, 10, out_ids[256:512].data_ptr(), out_sc[256:512].data_ptr(), out_cnt[256:512].data_ptr(), out_st[256:512].data_ptr(), 0)
    torch.cuda.synchronize()
    ids = out_ids[:512].cpu().numpy()
    rec = np.mean([len(set(ids[i].tolist()) & set(gt[i].tolist())) / 10 for i in range(512)])
    ix.enable_timing(True)
    ix.batch_search_device(Q[:256].data_ptr(), 256, 10, out_ids[:256].data_ptr(), out_sc[:256].data_ptr(), out_cnt[:256].data_ptr(), out_st[:256].data_ptr(), 0)
    st = ix.last_stats(0)
    ix.enable_timing(False)
    bytes_ = st.evals * (768 + 4) + st.adj_bytes
    print(f"ef={ef} recall@10={rec:.3f} evals/q={st.evals/256:.0f} exp/q={st.expansions/256:.0f} walk_ms={st.walk_ms:.2f} fin_ms={st.finalize_ms:.2f} prep_ms={st.prep_ms:.3f} GB/s(one batch)={bytes_/st.walk_ms/1e6:.1f}")
