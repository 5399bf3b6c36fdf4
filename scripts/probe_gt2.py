import sys; sys.path.insert(0,'/root/repo')
import numpy as np, cosdata_amd as ca
rng=np.random.default_rng(0)
for n,d,B in [(5000,1024,256),(5000,768,256),(300000,1024,256),(300000,1024,64),(300000,512,256)]:
    X=rng.standard_normal((n,d)).astype(np.float32); X/=np.linalg.norm(X,axis=1,keepdims=True)
    Q=X[rng.integers(0,n,B)]+0.05*rng.standard_normal((B,d)).astype(np.float32)
    ix=ca.HNSWIndex(d); ix.upload_vectors(X)
    ids,sc=ix.bruteforce_topk(Q,10)
    S=Q@X.T; gt=np.argsort(-S,axis=1)[:,:10]
    print(n,d,B, np.mean([len(set(ids[i])&set(gt[i]))/10 for i in range(B)]), sc[0][:3])
