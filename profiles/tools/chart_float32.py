"""Why the kernel runs the canonical chart's recursion in SQUARE-ROOT form (rl_on_manifold_amd/csrc/atacom_chart.h):
numpy emulation of both forms in float32 and float64 against the float64 specification (oracle/canonical_chart.py), on
the iiwa systems of tests/chart_cases.py that have joint-only charts and no stiff row (the part both emulations cover).

    python profiles/tools/chart_float32.py > profiles/r03_chart_float32.md
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from chart_cases import rollout_systems
from oracle import canonical_chart as cc
sy=rollout_systems('iiwa'); spec=sy['spec']
A_full,s,y=sy['A'][:6000],sy['s'][:6000],sy['y'][:6000]
n=len(A_full); nf=1; nq=6; k=5
rng=np.random.default_rng(4); alpha=rng.uniform(-10,10,(n,k))
info={}
ref=cc.canonical_mu(A_full,s,y,alpha,0.05,nf,info=info)
# restrict: no stiff rows, joint-only charts
arow=np.abs(A_full[:,1:,:]).max(2)
ok=(np.abs(s)>=0.03*arow).all(1) & (info['n_slack']==0)
print('# Canonical chart: covariance form vs square-root form of the same recursion, float32 vs float64\n')
print('%d iiwa systems (joint-only charts, no stiff row); error = max |mu - mu_spec| / max(1, |mu_spec|)\n' % ok.sum())
print('| form | precision | median | p90 | p99 | p99.9 | max |')
print('|---|---|---|---|---|---|---|')
def run(dtype, form):
    f=dtype
    A=A_full[ok,1:,:].astype(f); a=A_full[ok,0,:].astype(f); ss=s[ok].astype(f); yy=y[ok].astype(f); al=alpha[ok].astype(f)
    m=len(A); tol2=f(0.05*0.05)
    om=(f(1)/(ss*ss)).astype(f)
    M=(np.eye(nq,dtype=f)[None]+np.einsum('bg,bgi,bgj->bij',om,A,A)).astype(f)
    b=np.einsum('bg,bgi,bg->bi',om,A,yy[:,1:]).astype(f)
    C=np.linalg.cholesky(M).astype(f)
    Li=np.linalg.inv(C).astype(f)
    if form=='cov':
        G=np.einsum('bki,bkj->bij',Li,Li).astype(f)
        x=(-np.einsum('bij,bj->bi',G,b)).astype(f)
        t=np.einsum('bij,bj->bi',G,a).astype(f); S=(a*t).sum(1).astype(f)
        e=(-yy[:,0]-(a*x).sum(1)).astype(f)
        x=(x+t*(e/S)[:,None]).astype(f); G=(G-t[:,:,None]*t[:,None,:]/S[:,None,None]).astype(f)
        U=np.zeros((m,nq),f); nacc=np.zeros(m,int)
        for j in range(nq):
            dj=G[:,j,j]; acc=(nacc<k)&(dj>tol2)
            tv=al[np.arange(m),np.minimum(nacc,k-1)]
            inv=np.where(acc,f(1)/np.where(acc,dj,1),0).astype(f)
            col=G[:,:,j].copy(); coef=((tv-U[:,j])*inv).astype(f)
            U=(U+col*coef[:,None]).astype(f)
            G=(G-(col*inv[:,None])[:,:,None]*col[:,None,:]).astype(f)
            nacc+=acc
    else:
        V=np.swapaxes(Li,1,2).copy()   # V[:, i, :] = v_i = column i of Li  (R^T e_i)
        x=(-np.einsum('bki,bk->bi',Li,np.einsum('bij,bj->bi',Li,b))).astype(f)
        w=np.einsum('bi,bij->bj',a,V).astype(f)     # sum_i a_i v_i
        S=(w*w).sum(1).astype(f)
        g=np.einsum('bij,bj->bi',V,w).astype(f)
        e=(-yy[:,0]-(a*x).sum(1)).astype(f)
        x=(x+g*(e/S)[:,None]).astype(f)
        V=(V-(g/S[:,None])[:,:,None]*w[:,None,:]).astype(f)
        U=np.zeros((m,nq),f); nacc=np.zeros(m,int)
        for j in range(nq):
            w=V[:,j,:].copy(); dj=(w*w).sum(1).astype(f); acc=(nacc<k)&(dj>tol2)
            tv=al[np.arange(m),np.minimum(nacc,k-1)]
            inv=np.where(acc,f(1)/np.where(acc,dj,1),0).astype(f)
            g=np.einsum('bij,bj->bi',V,w).astype(f)
            coef=((tv-U[:,j])*inv).astype(f)
            U=(U+g*coef[:,None]).astype(f)
            V=(V-(g*inv[:,None])[:,:,None]*w[:,None,:]).astype(f)
            nacc+=acc
    aa=(a*a).sum(1); 
    U=U-a*((a*U).sum(1)/aa)[:,None]; x=x-a*(((a*x).sum(1)+yy[:,0])/aa)[:,None]
    u=(x+U).astype(f)
    wv=(-(yy[:,1:]+np.einsum('bgi,bi->bg',A,u))/ss).astype(f)
    return np.concatenate([u,wv],1).astype(np.float64)
r=ref[ok]; sc=np.maximum(1,np.abs(r).max(1))
for form in ('cov','sqrt'):
    for dt in (np.float64,np.float32):
        out=run(dt,form); err=np.abs(out-r).max(1)/sc
        print('| %s | %s | %.2e | %.2e | %.2e | %.2e | %.2e |' % ({'cov': 'covariance (Gamma, pivots off its diagonal)', 'sqrt': 'square root (vectors v_i, pivots = squared norms)'}[form], dt.__name__, np.median(err), np.quantile(err,.9), np.quantile(err,.99), np.quantile(err,.999), err.max()))

print()
print('A pivot near the tolerance (tol^2 = 2.5e-3) is what is left of O(1) entries after up to five rank-one downdates: in the')
print('covariance form its relative error is eps / tol^2 (and one sample in 6000 flips a decision), in the square-root form')
print('eps / tol.  Measured on the GPU (profiles/tools/gpu_chart_probe.py, 6000 iiwa systems incl. slack charts and stiff rows, float32')
print('kernel vs float64 specification): covariance form median 4.3e-6 / p99 3.5e-3 / max 0.98; square-root form median 1.4e-7 /')
print('p99 4.2e-6 / max 9.5e-5.')
