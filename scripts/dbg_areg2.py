import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import cosdata_amd as ca
import helpers as H
from oracle import oracle as O
n, dim, B = 40000, 128, 64
X = H.clustered_corpus(n, dim, n_centers=40, sigma=0.2, seed=23)
Q = H.queries_from(X, B, noise=0.05, seed=8)
ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_SUBBYTE), 2))
ix.upload_vectors(X)
os.environ["COS_FLAT_TILE_KERNEL"] = "1"
b = ix.flat_search(Q, 10)
os.environ.pop("COS_FLAT_TILE_KERNEL", None)
for grid in ["100000", "256", "128", "1"]:
    os.environ["COS_AREG_GRID"] = grid
    for rep in range(2):
        a = ix.flat_search(Q, 10)
        bad = [i for i in range(B) if not np.array_equal(a[0][i], b[0][i])]
        miss = sorted(set(int(x) for i in bad for x in b[0][i] if x not in a[0][i]))
        extra = sorted(set(int(x) for i in bad for x in a[0][i] if x not in b[0][i]))
        print("grid", grid, "bad queries:", bad[:20], "missing:", miss[:12], "extra:", extra[:12])
