"""Experiment driver (not part of the product; CPU only): counts the load instructions the streaming RoIAlign kernel issues per RoI on the
bench's proposals (gpurun_out/rois.pt, written by scripts/dump_rois.py on a GPU box), against exact-length streams and against every
footprint pixel loaded once. Round 6: the pipelined loops' prefetches past the end of each stream (D groups + rounding to whole D-groups,
seven streams per RoI, 6-7 wave steps per stream in the median) are 56 % of all load instructions -- not the re-walk of the footprint
per inner bin, which costs (L + 7) / L ~ 1.3x. The simulated 24.5 GB of loads + 1.7 GB of stores reproduces the measured
TCP_TOTAL_CACHE_ACCESSES (27.1 GB, profiles/r05_roi_align_roof.txt)."""
import torch, math, numpy as np
import os
d = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'rois.pt'))
boxes, bidx, counts = d['boxes'], d['batch_idx'], d['counts']
boxes = boxes.reshape(-1,4); bidx=bidx.reshape(-1)
real = bidx>=0
b = boxes[real].numpy().astype(np.float32)
print(b.shape)
sz = np.sqrt((b[:,2]-b[:,0])*(b[:,3]-b[:,1]))
lv = np.floor(4+np.log2(sz/224+1e-8)).clip(2,5).astype(int)-2
scale = np.array([1/4,1/8,1/16,1/32],dtype=np.float32)[lv]
H = np.array([200,100,50,25])[lv]; W=np.array([336,168,84,42])[lv]
def axis(start,end,size,P=7):
    # returns per-bin lo, n
    r = end-start; bs = r/P; g = np.ceil(r/P).astype(int).clip(1)
    los=[];ns=[]
    for bn in range(P):
        first=np.full(len(start),10**9); last=np.full(len(start),-1)
        gmax=g.max()
        for i in range(gmax):
            v = start+bn*bs+(i+.5)*bs/g
            ok = (i<g)&(v>=-1)&(v<=size)
            v2 = np.maximum(v,0)
            l = np.floor(v2).astype(int); h=l+1
            cl = l>=size-1
            l=np.where(cl,size-1,l); h=np.where(cl,size-1,h)
            first=np.where(ok,np.minimum(first,l),first); last=np.where(ok,np.maximum(last,h),last)
        n=np.where(last>=0,last-first+1,0); los.append(np.where(last>=0,first,0)); ns.append(n)
    return np.stack(los,1),np.stack(ns,1),g
sw=b[:,0]*scale-.5; ew=b[:,2]*scale-.5; sh=b[:,1]*scale-.5; eh=b[:,3]*scale-.5
lox,nx,gw=axis(sw,ew,W); loy,ny,gh=axis(sh,eh,H)
def ext(lo,n):
    l=np.where(n>0,lo,10**9).min(1); h=np.where(n>0,lo+n,0).max(1); return np.where(l<10**9,h-l,0)
ex,ey=ext(lox,nx),ext(loy,ny)
print('mean ext x,y',ex.mean(),ey.mean(),'footprint px mean',(ex*ey).mean(),'total Mpx',(ex*ey).sum()/1e6)
for q in (10,25,50,75,90,99): print(q, np.percentile(ex,q),np.percentile(ey,q),np.percentile(ex*ey,q))
# current kernel: outer = shorter side
outer_is_x = ex<=ey
nstep=np.where(outer_is_x,ex,ey); ni=np.where(outer_is_x[:,None],ny,nx)  # inner bins' n
def cur_loads(nstep,ni):
    tot=np.zeros(len(nstep))
    for pb in range(7):
        n=ni[:,pb]
        pair=(n>=1)&(n<=4)
        NY=np.clip(n,1,4)
        D=np.select([NY==1,NY==2,NY==3],[6,4,3],2)
        nws=(nstep+1)//2
        iters=np.ceil(nws/D)
        wl_pair=(iters*D+D)*NY  # wave loads (each 2 px)
        # non-pair path: NY<=6: D = 8,8,6,4,4,3
        NY2=np.clip(n,1,6); D2=np.select([NY2<=2,NY2==3,NY2<=5],[8,6,4],3)
        it2=np.ceil(nstep/D2); wl2=(it2*D2+D2)*NY2
        tall = n>6
        wl3 = nstep*np.ceil(n/6)*6
        tot+=np.where(pair,wl_pair,np.where(tall,wl3,wl2))
    return tot
cl=cur_loads(nstep,ni)
stores=28
print('current: wave-loads/RoI mean',cl.mean(),' total wave instr (M)',(cl+stores).sum()/1e6, 'bytes thru L1 (GB) approx', (cl*1024).sum()/1e9)
ideal=(ex*ey)/2
print('ideal pair loads/RoI',ideal.mean(), 'total (M)',(ideal+25).sum()/1e6)
# exact-length (no overshoot) pair path
def exact(nstep,ni):
    tot=np.zeros(len(nstep))
    for pb in range(7):
        n=ni[:,pb]; nws=(nstep+1)//2
        tot+=nws*n
    return tot
el=exact(nstep,ni); print('exact-length loads/RoI',el.mean(),'total (M)',(el+28).sum()/1e6)
