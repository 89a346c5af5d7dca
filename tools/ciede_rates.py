"""Decision rate of the branch-free CIEDE2000 pass (nq_selftest_ciede) on random Lab pairs and on close neighbours; equality with the
literal functions among decided AND undecided pairs.  GPU box: python tools/ciede_rates.py"""
import sys; sys.path.insert(0,'.')
import numpy as np, nquant.android_amd as nq
rng=np.random.default_rng(1); n=400000
def lab(n,ch=128.0): return np.stack([rng.uniform(0,100,n),rng.uniform(-ch,ch,n),rng.uniform(-ch,ch,n)],axis=1)
q=nq.PnnLABQuantizer(np.zeros((1,1),np.int32))
a=lab(n)
for name,s in [("any",np.concatenate([lab(n),lab(n)],1)),("grey2",np.concatenate([lab(n,2),lab(n,2)],1))]+[("pert %g"%sg,np.concatenate([a,a+rng.normal(0,sg,a.shape)],1)) for sg in (5,0.5,0.05,1e-3)]:
    f,l,ok=q.selftest_ciede(s.astype(np.float32)); dec=ok==1
    print(name,"decided %.4f"%dec.mean(),"mismatch among decided",int((f[dec]!=l[dec]).any(1).sum()), "fast==lit among undecided %.3f"%((f[~dec]==l[~dec]).all(1).mean() if (~dec).any() else 1))
