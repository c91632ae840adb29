import sys; sys.path.insert(0,".")
import numpy as np
from trlda_amd import _ffi
from trlda_amd.models import OnlineLDA
from trlda_amd.documents import CSRDocuments
from oracle.pyoracle import Oracle
o=Oracle(); L=_ffi.lib()
K,V=100,7000
rng=np.random.RandomState(0)
for n in (128,129,130,137,150,192,193):
    ids=rng.permutation(V)[:n].astype(np.int32); cnts=(1+rng.randint(3,size=n)).astype(np.int32)
    ip=np.array([0,n],np.int32)
    o.seed(1); lam=o.sample_gamma(K,V,3)/3.; o.seed(2); g0=o.sample_gamma(K,1,100)/100.
    m=OnlineLDA(V,K,10); m.lambdas=lam
    for it in (0,1,5):
        g,s,_=m.update_variables(CSRDocuments(ip,ids,cnts),latents=g0,max_iter=it,threshold=0.,return_iterations=True)
        go,so,_=o.estep(lam,.1,ip,ids,cnts,g0,it,0.)
        colerr=np.abs(s[:,ids]-so[:,ids]).max(axis=0)/np.abs(so[:,ids]).max(axis=0)
        print(n,it,"gamma err %.2e"%np.max(np.abs(g-go)/np.abs(go)),"sum gamma %.6f vs %.6f"%(g.sum(),go.sum()), "sstats worst cols", np.argsort(-colerr)[:3], colerr.max())
