"""Lane statistics of the edge-cost kernel's obstacle scans on the benchmark scenes, on the CPU (development aid; the numbers
quoted in HISTORY.md 3.1 and profiles/r02_edge/README.md): scans per edge, soft samples per scan, wave-level scans of the
kernel's lane mapping against a one-scene-per-wavefront mapping, the exact box test.  Usage: python tools/edge_lane_sim.py [scenes]"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from emplanner_carla_amd import scenes as S
cfg=S.CFG2
B=int(sys.argv[1]) if len(sys.argv)>1 else 700
b=S.make_batch(range(B),cfg)
row,col=cfg.row,cfg.col; rr=row*row
ss,sl=cfg.sample_s,cfg.sample_l
# pair table: l samples
def lat(i): return ((row+1)/2-1-i)*sl
t=np.arange(10)*ss/10
L=np.zeros((row,row,10))
for k in range(row):
    for i in range(row):
        l0,l1=lat(k),lat(i); h=l1-l0; T=ss
        a3=20*h/(2*T**3); a4=-30*h/(2*T**4); a5=12*h/(2*T**5)
        L[k,i]=l0+a3*t**3+a4*t**4+a5*t**5
llo=np.minimum.outer(np.array([lat(k) for k in range(row)]),np.array([lat(i) for i in range(row)]))
lhi=np.maximum.outer(np.array([lat(k) for k in range(row)]),np.array([lat(i) for i in range(row)]))
ps=b.sl_start[:,0]
J=np.arange(1,col)
s0=ps[:,None]+J[None,:]*ss               # B, 39
os_=b.sl_obs_s; ol_=b.sl_obs_l            # B, 8
near_s=(os_[:,None,:]>s0[:,:,None]-6.5)&(os_[:,None,:]<s0[:,:,None]+t[9]+6.5)    # B,39,8
latok=(ol_[:,:,None,None]>llo[None,None]-6.5)&(ol_[:,:,None,None]<lhi[None,None]+6.5)   # B,8,row,row
print("near obstacles per (scene,col): mean", near_s.sum(-1).mean(), "max", near_s.sum(-1).max())
print("lat pass frac", latok.mean())
# full pair activity: B,39,8,k,i
act=near_s[:,:,:,None,None]&latok[:,None,:,:,:]
print("scans per edge (lane-scans):", act.sum()/(B*39*rr))
# sample-level stats
sn=s0[:,:,None]+t[None,None,:]                                  # B,39,10
dlon=os_[:,None,:,None]-sn[:,:,None,:]                          # B,39,8,10
dlat=ol_[:,:,None,None,None]-L[None,None]                       # B,8,k,i,10
d2=dlon[:,:,:,None,None,:]**2+dlat[:,None,:,:,:,:]**2           # B,39,8,k,i,10
hard=d2<=16; soft=(d2<36)&~hard
firsthard=np.where(hard.any(-1),hard.argmax(-1),10)            # B,39,8,k,i
alive=np.arange(10)[None,None,None,None,None,:]<firsthard[...,None]
softc=(soft&alive)
print("per active scan: soft samples mean", softc.sum(-1)[act].mean(), " hard frac", (firsthard<10)[act].mean(), " mean evaluated samples", np.minimum(firsthard+1,10)[act].mean())
print("scans with zero soft and no hard:", ((softc.sum(-1)==0)&(firsthard==10))[act].mean())
Sx=64//row
tiles=B//Sx
# (a) current mapping: wave=(tile,j), loop k; iterations=max_s popcount(near_s); lanes active = act
it_a=0; lanes_a=0
ns=near_s[:tiles*Sx].reshape(tiles,Sx,39,8)
pop=ns.sum(-1)                      # tiles,Sx,39
it_a=pop.max(1).sum()*row           # per k
lanes_a=act[:tiles*Sx].sum()
print("(a) wave-scans", it_a, "lane-scans", lanes_a, "util", lanes_a/(it_a*64))
# (b) design X: items (s, jj, p) flattened per block of 8 columns; wave=64 consecutive; iterations = max over lanes popcount(near_s & lat)
cols_chunks=[(1+c*8, min(col,1+c*8+8)) for c in range(5)]
it_b=0; it_b2=0
for tl in range(tiles):
    for (ja,jb) in cols_chunks:
        # mask count per item
        a=act[tl*Sx:(tl+1)*Sx, ja-1:jb-1]     # Sx, nc, 8, k, i
        cnt=a.sum(2).reshape(-1)               # flattened (s, jj, k, i)
        nsr=np.repeat(near_s[tl*Sx:(tl+1)*Sx, ja-1:jb-1].sum(-1).reshape(-1), rr)
        n=len(cnt); pad=(-n)%64
        c2=np.concatenate([cnt,np.zeros(pad,int)]).reshape(-1,64)
        n2=np.concatenate([nsr,np.zeros(pad,int)]).reshape(-1,64)
        it_b+=c2.max(1).sum(); it_b2+=n2.max(1).sum()
print("(b) wave-scans (mask incl. lat)", it_b, "util", lanes_a/(it_b*64), "; (lon-only mask)", it_b2, lanes_a/(it_b2*64))
# pruning options
s9=s0+t[9]
dx=np.maximum(np.maximum(s0[:,:,None]-os_[:,None,:], os_[:,None,:]-s9[:,:,None]),0)     # B,39,8
dy=np.maximum(np.maximum(llo[None,None]-ol_[:,:,None,None], ol_[:,:,None,None]-lhi[None,None]),0)   # B,8,k,i
box=(dx[:,:,:,None,None]**2+dy[:,None,:,:,:]**2)<36.0
print("box-test scans per edge:", box.sum()/(B*39*rr), " vs axis test", act.sum()/(B*39*rr))
useful=(d2<36).any(-1)
print("truly useful scans per edge:", useful.sum()/(B*39*rr))
print("useful & hard", (useful&(firsthard<10)).sum()/(B*39*rr))
# soft samples per useful scan
print("soft per useful scan", softc.sum(-1)[useful].mean(), "; samples with d2<36 (incl after hard)", (d2<36).sum(-1)[useful].mean())
# per wave (design X, box mask) utilisation
it=0
for tl in range(tiles):
    for (ja,jb) in cols_chunks:
        a=box[tl*Sx:(tl+1)*Sx, ja-1:jb-1]
        cnt=a.sum(2).reshape(-1); n=len(cnt); pad=(-n)%64
        it+=np.concatenate([cnt,np.zeros(pad,int)]).reshape(-1,64).max(1).sum()
print("design X with box mask: wave-scans", it, "util", box[:tiles*Sx].sum()/(it*64), " vs current wave-scans", it_a)
