"""Two ways of carrying a label from one tree level to the next in the multiscale Gibbs product sampler, in plain numpy (no
library code), on the door scenario of BASELINE config 3: density A = an odometry-propagated belief (sigma 0.25 around the
truth) of which a share sits one door spacing (1.6) off, density B = a four-mode sighting.  "oracle" = rounds 1-3's hand-down
(the label goes to one of the selected node's two children, drawn by their share of the leaves); "ihler" = the published
algorithm (Ihler et al., NIPS 2003; KernelDensityEstimate.jl's gibbs1): a point x from the selected Gaussians, every label
re-drawn on x among ALL nodes of the level.  One sweep per level either way.  Printed: the share of the product's samples in
the wrong mode against the exact product of the two kernel density estimates.  -> profiles/r04_sampler_variants.txt"""
import numpy as np
rng=np.random.default_rng(0)
def build(x,h):
    """levels of a balanced KD tree over sorted 1-D points: list of (mean,var,weight) arrays per level"""
    xs=np.sort(x); N=len(xs)
    segs=[(0,N)]; levels=[]
    while True:
        m=np.array([xs[a:b].mean() for a,b in segs]); v=np.array([xs[a:b].var() for a,b in segs])+h*h
        w=np.array([(b-a)/N for a,b in segs]); levels.append((m,v,w,list(segs)))
        if all(b-a<=1 for a,b in segs): break
        new=[]
        for a,b in segs:
            if b-a<=1: new.append((a,b))
            else:
                mid=a+(b-a+1)//2; new+= [(a,mid),(mid,b)]
        segs=new
    return levels
def children(levels,l,z):
    """indices at level l+1 of node z's children"""
    segs=levels[l][3]; nxt=levels[l+1][3]
    a,b=segs[z]
    return [k for k,(c,d) in enumerate(nxt) if c>=a and d<=b]
def pick(logw):
    w=np.exp(logw-logw.max()); w/=w.sum(); return rng.choice(len(w),p=w)
def cond(levels_j,l,mu,var):
    m,v,w,_=levels_j[l]
    return -0.5*((m-mu)**2/(v+var)+np.log(v+var))+np.log(w)
def sample(trees,variant,niter=1):
    F=len(trees); L=len(trees[0])-1
    ind=[0]*F
    for l in range(1,L+1):
        if variant=="oracle":
            for j in range(F):
                ch=children(trees[j],l-1,ind[j])
                if len(ch)==1: ind[j]=ch[0]
                else:
                    nl=trees[j][l][2][ch[0]]; nr=trees[j][l][2][ch[1]]
                    ind[j]=ch[0] if rng.random()*(nl+nr)<nl else ch[1]
        else:  # ihler: x from product of the selected nodes, then every label given x among ALL nodes of level l
            prec=sum(1/trees[j][l-1][1][ind[j]] for j in range(F)); mu=sum(trees[j][l-1][0][ind[j]]/trees[j][l-1][1][ind[j]] for j in range(F))/prec
            x=mu+rng.standard_normal()/np.sqrt(prec)
            for j in range(F):
                ind[j]=pick(cond(trees[j],l,x,0.0))
        for it in range(niter):
            for j in range(F):
                prec=sum(1/trees[q][l][1][ind[q]] for q in range(F) if q!=j)
                mu=sum(trees[q][l][0][ind[q]]/trees[q][l][1][ind[q]] for q in range(F) if q!=j)/prec
                ind[j]=pick(cond(trees[j],l,mu,1/prec))
    prec=sum(1/trees[j][L][1][ind[j]] for j in range(F)); mu=sum(trees[j][L][0][ind[j]]/trees[j][L][1][ind[j]] for j in range(F))/prec
    return mu+rng.standard_normal()/np.sqrt(prec)
def exact_share(A,hA,B,hB,lo,hi):
    g=np.linspace(-4,5,9001)
    pa=np.exp(-0.5*((g[:,None]-A[None,:])/hA)**2).sum(1); pb=np.exp(-0.5*((g[:,None]-B[None,:])/hB)**2).sum(1)
    p=pa*pb; p/=p.sum(); return p[(g>lo)&(g<hi)].sum()
import sys
N=200
for frac in (0.0,0.02,0.045,0.1):
    res={}
    for variant in ("oracle","ihler"):
        tot=[];ex=[]
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
            nw=int(round(frac*N))
            A=np.concatenate([0.25*rng.standard_normal(N-nw), 1.6+0.25*rng.standard_normal(nw)])
            B=np.concatenate([c+0.1*rng.standard_normal(N//4) for c in (-1.6,0,1.6,3.2)])
            hA,hB=0.09,0.04
            tA,tB=build(A,hA),build(B,hB)
            s=np.array([sample([tA,tB],variant) for _ in range(N)])
            tot.append(np.mean(np.abs(s-1.6)<0.8)); ex.append(exact_share(A,hA,B,hB,0.8,2.4))
        res[variant]=(np.mean(tot),np.mean(ex))
    print(f"wrong-mode share in A {frac:.3f}: exact product {res['oracle'][1]:.3f}  oracle-variant {res['oracle'][0]:.3f}  ihler-variant {res['ihler'][0]:.3f}",flush=True)
