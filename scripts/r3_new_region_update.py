# experiment: incremental update that adds items of NEW clusters to an existing index
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import hannoy_amd as hny
from oracle import orc
from conftest import draw_levels
rng=np.random.default_rng(5)
dim=48; nA, nB = 30000, 12000
centA=rng.uniform(-1,1,(30,dim)).astype(np.float32); centB=rng.uniform(-1,1,(12,dim)).astype(np.float32)
A=(centA[rng.integers(0,30,nA)]+0.15*rng.standard_normal((nA,dim))).astype(np.float32)
B=(centB[rng.integers(0,12,nB)]+0.15*rng.standard_normal((nB,dim))).astype(np.float32)
qs=(centB[rng.integers(0,12,300)]+0.15*rng.standard_normal((300,dim))).astype(np.float32)
allv=np.concatenate([A,B]); ids=np.arange(nA+nB,dtype=np.uint32)
d2=((qs**2).sum(1)[:,None]-2*qs@allv.T+(allv**2).sum(1)[None,:]); truth=np.argsort(d2,axis=1)[:,:10]
kw=dict(M=16,M0=32,ef_construction=64)
dsA=orc.Dataset.from_f32(1,A,draw_levels(nA,16,1)); itA=hny.ItemSet(1,dim,dsA.ids,dsA.codes,dsA.headers,dsA.levels)
gA=hny.build(itA,**kw)
dsAll=orc.Dataset.from_f32(1,allv,np.zeros(nA+nB,np.uint8)); lvB=draw_levels(nB,16,2)
itAll=hny.ItemSet(1,dim,dsAll.ids,dsAll.codes,dsAll.headers,lvB)
qc=orc.encode_vectors(1,qs); qh=orc.make_headers(1,dim,qc)
def rec(g, items):
    with hny.Builder(items, prev=g, load=True, **kw) as b:
        i,_,c=b.search_knn(qc,qh,k=10,ef_search=64)
    return sum(len(set(i[k,:c[k]].tolist())&set(truth[k].tolist())) for k in range(len(qs)))/truth.size
for bm in (0, 1024, 64):
    g2=hny.build_incremental(itAll,gA,np.arange(nA,nA+nB,dtype=np.uint32),[],batch_max=bm,**kw)
    print("incremental add of new clusters, batch_max",bm,"batches",g2.n_batches,"recall@10 on the new region",round(rec(g2,itAll),4))
lv=np.concatenate([dsA.levels,lvB]); dsF=orc.Dataset.from_f32(1,allv,lv); itF=hny.ItemSet(1,dim,dsF.ids,dsF.codes,dsF.headers,lv)
gF=hny.build(itF,**kw); print("fresh build of everything: recall",round(rec(gF,itF),4))
