"""Rounding-level difference between Gauss-Jordan (same pivots) and LINPACK-style LU + back substitution
on the rate matrices of BASELINE config 2 (numpy, CPU; uses the oracle only to assemble matrices)."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.molecule import SYNTH_CO_PATH
mol = O.Molecule(SYNTH_CO_PATH); cfg = workloads.config2(1024); n=41
def gepp(A):
    A=A.copy(); b=np.zeros(n); b[n-1]=1.0
    for k in range(n-1):
        l=k+np.argmax(np.abs(A[k:,k]))
        if l!=k: A[[k,l]]=A[[l,k]]; b[[k,l]]=b[[l,k]]
        t=-1.0/A[k,k]; m=A[k+1:,k]*t
        A[k+1:,k+1:]+=np.outer(m,A[k,k+1:]); b[k+1:]+=m*b[k]
    x=np.zeros(n)
    for k in range(n-1,-1,-1):
        x[k]=b[k]/A[k,k]; b[:k]-=x[k]*A[:k,k]
    return x
def gj(A):
    A=A.copy(); b=np.zeros(n); b[n-1]=1.0
    for k in range(n):
        l=k+np.argmax(np.abs(A[k:,k]))
        if l!=k: A[[k,l]]=A[[l,k]]; b[[k,l]]=b[[l,k]]
        t=-1.0/A[k,k]; m=A[:,k]*t; m[k]=0.0
        A[:,k+1:]+=np.outer(m,A[k,k+1:]); b+=m*b[k]
    return b/np.diag(A)
worst=0; worst_big=0
for w in range(0,1024,16):
    p = cfg['walkers'][w]
    st = O.State(mol, 2, 1.0); st.backrad(cfg['tbg'])
    nn = 10**p[0]; st.set_density({2:0.25*nn, 3:0.75*nn}); st.s.tkin = 10**p[1]; st.s.cdmol = 10**p[2]; st.rates()
    conv=0; it=0
    while not conv and it<12:
        conv=st.matrix(it); it+=1
        Y=np.ctypeslib.as_array(st.s.yrate,(n*n,)).copy().reshape(n,n,order='F')
        Y[n-1,:]=1.0
        x1=gepp(Y); x2=gj(Y)
        tot1=x1.sum()
        if not np.isfinite(tot1) or not np.all(np.isfinite(x2)): continue
        rel=np.abs(x1-x2)/np.maximum(np.abs(x1),1e-300)
        big=x1/tot1>1e-8
        worst=max(worst, rel.max()); worst_big=max(worst_big, rel[big].max())
print("max rel diff GJ vs GEPP: all levels %.2e ; levels with x>1e-8: %.2e" % (worst, worst_big))
