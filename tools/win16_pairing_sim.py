"""How many strip steps the packed region windows (c4_win16_kernel.h) spend on the north-star batch under three pairings of
the window chains: each job alone, the host-made pairs that stay together for a whole chain (round 5), pairs made anew at
every hop (the experiment of profiles/r06_hopsync_experiment.md).  The planted structure of workloads.est2genome_pairs stands in for the paths (no DP).
Output on the north-star batch: 47 036 strip steps per pair (host pairs, lane use 0.68) against 39 031 (paired per hop, 0.82)."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
# reproduce the planted structure of est2genome_pairs without building sequences
def structure(k, qlen=1000, tlen=100000, seed=20260932):
    rng = np.random.default_rng([seed, k])
    q = rng.integers(0,4,size=qlen)
    ncut = int(rng.integers(3,7))
    cuts = np.sort(rng.choice(np.arange(30,qlen-30), size=ncut, replace=False))
    # replicate random consumption roughly: need exact? we only need statistics -> draw independently
    ilens=[int(rng.integers(100,5001)) for _ in range(ncut)]
    gene_len = qlen + sum(ilens)
    flank = tlen-gene_len
    left = int(rng.integers(0,flank+1))
    return cuts, ilens, left
K=4096; DC=2
def chain(cuts, ilens, left, qlen=1000):
    # path: target col of query row r: left + r + sum(ilens for cuts<r)
    def col_of_row(r):
        return left + r + sum(l for c,l in zip(cuts,ilens) if c<r)
    def row_of_col(c):
        # largest row r with col_of_row(r) <= c  (inside intron -> row = cut)
        lo,hi=0,qlen
        while lo<hi:
            m=(lo+hi+1)//2
            if col_of_row(m)<=c: lo=m
            else: hi=m-1
        return lo
    te=col_of_row(qlen); ts=left
    hops=[]
    d=te>>12  # dump index left of te: window starts at d*K-(DC-1)
    Q=qlen; endcol=te
    while True:
        t0w = d*K-(DC-1) if d>=1 else 0
        T=endcol-t0w
        hops.append((Q,T))
        if t0w<=ts or d==0: break
        # entry at column t0w(ish): row there
        r=row_of_col(t0w)
        Q=r; endcol=t0w+1; d-=1
        if Q<=0: break
    return hops
rng_all=[structure(k) for k in range(4096)]
chains=[chain(*s) for s in rng_all]
nh=np.array([len(c) for c in chains]); print('hops mean',nh.mean(),'max',nh.max())
R=4;W=64*R
def steps(Q,T): return ((Q+1+W-1)//W)*(T+64)
solo=sum(steps(Q,T) for c in chains for Q,T in c)
cells=sum((Q+1)*(T+1) for c in chains for Q,T in c)
print('solo strip-steps per job',solo/4096,'cells/job',cells/4096,'lane util solo',cells/(solo*W))
# current pairing: sort by first window cells desc, pair adjacent; chain in lockstep
order=sorted(range(4096),key=lambda i:-(chains[i][0][0]+1)*(chains[i][0][1]+1))
tot=0
for x in range(0,4096,2):
    a,b=chains[order[x]],chains[order[x+1]]
    for h in range(max(len(a),len(b))):
        Qa,Ta=a[h] if h<len(a) else (0,2)
        Qb,Tb=b[h] if h<len(b) else (0,2)
        tot+=steps(max(Qa,Qb),max(Ta,Tb))
print('paired strip-steps per pair',tot/2048,'util',cells/(tot*W*2))
# re-pair per hop: all hop-h windows sorted by (strips,T)
tot2=0
maxh=nh.max()
for h in range(maxh):
    wins=[c[h] for c in chains if h<len(c)]
    wins.sort(key=lambda w:(-((w[0]+1+W-1)//W),-w[1]))
    for x in range(0,len(wins),2):
        a=wins[x]; b=wins[x+1] if x+1<len(wins) else (0,2)
        tot2+=steps(max(a[0],b[0]),max(a[1],b[1]))
print('re-paired per hop strip-steps per pair',tot2/2048,'util',cells/(tot2*W*2))
for RR in (1,2,3):
    WW=64*RR
    s=sum(((Q+1+WW-1)//WW)*(T+64) for c in chains for Q,T in c)
    print('R',RR,'solo util',cells/(s*WW))
