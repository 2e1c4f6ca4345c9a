#!/usr/bin/env python3
"""Would a cell-ordered table + bounding-box culling make the exact k-NN scan sub-quadratic at BASELINE config 4's
shape (d = 8, k = 50, iid Gaussian points)?  A CPU simulation of the pruning rule a block-granular matrix-core scan
could use (round 6; the reference hands `algorithm=` to scikit-learn's kd / ball trees, src/MuyGPyS/neighbors.py:89-107):

  * order the table by a balanced kd tree down to leaves of one staged tile (128 rows), bounding box per tile;
  * a query block (QB consecutive rows of that order = what one workgroup owns) must scan every tile whose box-to-box
    lower bound is within the block's largest k-th neighbour distance (the scan tests 32 x 32 blocks of pairs on the
    matrix cores: it cannot prune per query).

Prints the fraction of tiles that survive, with the exact k-th distances and with the estimate a first pass over the
2 048 rows around the block would give.  Result (profiles/r06_knn_prune_sim.txt): 96-100 % of the tiles survive at
1 M points; at 10 M, 60 % with the exact radii and 91 % with the first-pass estimate -- in eight dimensions 13-17 tree
levels split each coordinate about twice, a tile's box spans a quarter to a half of the data's range in every
coordinate, and the block's worst query (a tail point of the Gaussian) sets the radius.  A 10-40 % saving does not pay
for the ordering, the first pass and the culling; the quadratic scan stays.

    python tools/knn_prune_sim.py [N] [QB]
"""
import numpy as np, torch, time, sys
N=int(sys.argv[1]) if len(sys.argv)>1 else 1_000_000
d,k=8,50; TN=128; QB=int(sys.argv[2]) if len(sys.argv)>2 else 256
rng=np.random.default_rng(0)
X=rng.standard_normal((N,d)).astype(np.float32)
t=time.time()
# kd ordering: at each level sort by (node, value[dim]); node boundaries fixed by position
perm=np.arange(N)
levels=int(np.ceil(np.log2(N/TN)))
node=np.zeros(N,dtype=np.int64)
for lv in range(levels):
    dim=lv%d
    v=X[perm,dim]
    order=np.lexsort((v,node))
    perm=perm[order]
    pos=np.arange(N)
    node=(pos*(2**(lv+1))//N)
print("kd order",time.time()-t,"levels",levels)
Xs=X[perm]
nt=(N+TN-1)//TN
pad=nt*TN-N
Xp=np.concatenate([Xs,np.repeat(Xs[-1:],pad,0)]) if pad else Xs
T=Xp.reshape(nt,TN,d); tlo=T.min(1); thi=T.max(1)
nb=(N+QB-1)//QB
Xt=torch.from_numpy(Xs)
sample=rng.choice(nb-1,size=48,replace=False)
fr=[];fr_win=[]
for bidx in sample:
    q=Xt[bidx*QB:(bidx+1)*QB]
    # exact kth distance
    d2=(q*q).sum(1)[:,None]+(Xt*Xt).sum(1)[None,:]-2*q@Xt.T
    kth=torch.topk(d2,k+1,dim=1,largest=False).values[:,-1].numpy()
    # window tau (2048 rows around)
    r0=max(0,bidx*QB-896); r1=min(N,r0+2048)
    d2w=d2[:,r0:r1]; kthw=torch.topk(d2w,k+1,dim=1,largest=False).values[:,-1].numpy()
    qlo=q.numpy().min(0); qhi=q.numpy().max(0)
    gap=np.maximum(0,np.maximum(tlo-qhi,qlo-thi)); lb2=(gap*gap).sum(1)
    fr.append((lb2<=kth.max()).mean()); fr_win.append((lb2<=kthw.max()).mean())
print(f"N={N} QB={QB}: tiles surviving with exact tau_B: mean {np.mean(fr):.4f} median {np.median(fr):.4f} max {np.max(fr):.4f}; with window tau: mean {np.mean(fr_win):.4f} median {np.median(fr_win):.4f}")
