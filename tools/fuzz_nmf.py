"""Randomised odd-input sweep of pymf_amd.NMF against the oracle (dtype, layout, m/n/k of 1, k > n ...).
ferr is compared relative to ||V||: an (almost) exact fit leaves a float32-sized residual floor."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pymf_amd
from oracle import NMFOracle, SNMFOracle
from pymf_amd.rnmf import RNMF
rs = np.random.RandomState(0)
def rel(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30)
bad = 0
cases = []
for trial in range(40):
    m = int(rs.choice([1,2,3,17,64,65,200,1000])); n = int(rs.choice([1,2,5,63,64,65,130,257,400,700]))
    k = int(rs.choice([1,2,3,16,17,33,64,65,100,128]))
    kind = rs.choice(["f32","f64","fortran","slice","int"])
    V = rs.random_sample((m,n))
    if kind=="f32": V=V.astype(np.float32)
    elif kind=="fortran": V=np.asfortranarray(V.astype(np.float32))
    elif kind=="slice": V=np.ascontiguousarray(rs.random_sample((2*m,2*n)).astype(np.float32))[::2, ::2]
    elif kind=="int": V=(V*10).astype(np.int64)
    W0=rs.random_sample((m,k)); H0=rs.random_sample((k,n))
    for cls,orc in ((pymf_amd.NMF,NMFOracle),):
        try:
            a=cls(V,num_bases=k); a.W,a.H=W0.copy(),H0.copy(); a.factorize(niter=3)
            o=orc(np.asarray(V,dtype=np.float64) if kind=="int" else V,num_bases=k); o.W,o.H=W0.copy(),H0.copy(); o.factorize(niter=3)
            e=max(rel(a.W,o.W),rel(a.H,o.H)); fe=abs(a.ferr[-1]-o.ferr[-1])/max(np.linalg.norm(np.asarray(V,dtype=np.float64)),1e-12)   # float32 floor: relative to ||V||
            flag = "" if (e<2e-5 and fe<2e-6) else "  <<<<<"
            if flag: bad+=1
            print(m,n,k,kind,cls.__name__,"relWH %.2e ferr %.2e"%(e,fe),flag)
        except Exception as ex:
            bad+=1; print(m,n,k,kind,cls.__name__,"EXC",type(ex).__name__,str(ex)[:100])
print("bad",bad)
