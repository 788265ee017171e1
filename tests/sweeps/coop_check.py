import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import pymf_amd
from pymf_amd import _lib
from oracle import NMFOracle, BNMFOracle
def rel(a,b): return np.linalg.norm(np.asarray(a,np.float64)-b)/np.linalg.norm(b)
for (m,n,k) in ((3000,256,128),(777,200,100),(64,64,65),(5000,128,128),(130,256,70),(1000,330,100),(1500,384,64),(800,500,33),(4000,512,64),(70,300,50)):
    rs=np.random.RandomState(m+k); V=rs.random_sample((m,n)).astype(np.float32)
    W0,H0=rs.random_sample((m,k)),rs.random_sample((k,n))
    a=pymf_amd.NMF(V,num_bases=k); a.W,a.H=W0.copy(),H0.copy(); a.factorize(niter=4)
    o=NMFOracle(V,num_bases=k); o.W,o.H=W0.copy(),H0.copy(); o.factorize(niter=4)
    print(m,n,k,a._ctx.path_name,"relW %.3g relH %.3g relferr %.3g"%(rel(a.W,o.W),rel(a.H,o.H),np.max(np.abs(a.ferr-o.ferr)/o.ferr)))
