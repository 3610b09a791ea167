"""Model of the generalised register merge of warp_splat_windows_kernel: counts what is left for LDS atomics."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_render import scene
h,w=704,1280
depth,img,K=scene(h,w)
f=np.float32
Kinv=np.linalg.inv(K.astype(np.float64)).astype(f)
ys,xs=np.mgrid[0:h,0:w].astype(f)
px=(Kinv[0,0]*xs+Kinv[0,1]*ys+Kinv[0,2]).astype(f); py=(Kinv[1,0]*xs+Kinv[1,1]*ys+Kinv[1,2]).astype(f)
X=(px*depth).astype(f); Y=(py*depth).astype(f); Z=depth
def geom(tx):
    cx=(X+f(tx)).astype(f)
    pr0=((K[0,0]*cx+K[0,1]*Y).astype(f)+K[0,2]*Z).astype(f); pr1=((K[1,0]*cx+K[1,1]*Y).astype(f)+K[1,2]*Z).astype(f)
    u=(pr0/(Z+f(1e-7))).astype(f); v=(pr1/(Z+f(1e-7))).astype(f)
    ox=((u-xs).astype(f)+xs+1).astype(f); oy=((v-ys).astype(f)+ys+1).astype(f)
    cl=lambda a,hi: np.clip(a,0,hi).astype(np.int64)
    return cl(np.floor(ox),w+1),cl(np.ceil(ox),w+1),cl(np.floor(oy),h+1),cl(np.ceil(oy),h+1)
def sim_tile(FX,CX,FY,CY,ty0,tx0, general=True, census=False):
    # returns (#store corners, #atomic corners) ; values modelled as 1.0 weights to check conservation
    tot_atomic=0; tot_store=0
    contrib={}  # texel -> total (for conservation)
    final={}
    owner={}
    pend=[]  # (tex, val, kind)
    log=[]
    for wv in range(4):
        # wave: lanes 0..63 ; thread t = wv*64+lane ; rows ty0 + 4*(t>>5) + k ; col tx0 + (t&31)
        TW=np.full((64,4,2),-1,np.int64); TE=np.full((64,4,2),-1,np.int64)
        W=np.zeros((64,4,2)); E=np.zeros((64,4,2))
        for lane in range(64):
            t=wv*64+lane
            for k in range(4):
                y=ty0+4*(t>>5)+k; x=tx0+(t&31)
                if y>=h or x>=w: continue
                fx,cx,fy,cy=FX[y,x],CX[y,x],FY[y,x],CY[y,x]
                TW[lane,k]=[fy*65536+fx, cy*65536+fx]; TE[lane,k]=[fy*65536+cx, cy*65536+cx]
                W[lane,k]=[1,1]; E[lane,k]=[1,1]
                for tx_ in (fy*65536+fx, cy*65536+fx, fy*65536+cx, cy*65536+cx): contrib[tx_]=contrib.get(tx_,0)+1
        if general:
            for lane in range(64):
                for k in range(4):
                    if TW[lane,k,0]<0: continue
                    if TW[lane,k,1]==TW[lane,k,0]: W[lane,k,0]+=W[lane,k,1]; TW[lane,k,1]=-1; W[lane,k,1]=0
                    if TE[lane,k,1]==TE[lane,k,0]: E[lane,k,0]+=E[lane,k,1]; TE[lane,k,1]=-1; E[lane,k,1]=0
                    for c in range(2):
                        if TE[lane,k,c]<0: continue
                        for cc in range(2):
                            if TW[lane,k,cc]==TE[lane,k,c]: W[lane,k,cc]+=E[lane,k,c]; TE[lane,k,c]=-1; E[lane,k,c]=0; break
            # horizontal: snapshot donors first (DPP reads pre-merge values of lane-1's EAST corners: east never receives, so fine)
            taken=np.zeros((64,4,2),bool)
            for lane in range(64):
                if lane&31==0: continue
                for k in range(4):
                    for c in range(2):
                        T=TE[lane-1,k,c]
                        if T<0: continue
                        done=False
                        for j in (k,k-1,k+1):
                            if j<0 or j>3 or done: continue
                            for cc in range(2):
                                if TW[lane,j,cc]==T:
                                    W[lane,j,cc]+=E[lane-1,k,c]; taken[lane-1,k,c]=True; done=True; break
            for lane in range(64):
                for k in range(4):
                    for c in range(2):
                        if taken[lane,k,c]: TE[lane,k,c]=-1; E[lane,k,c]=0
            # vertical in thread
            for lane in range(64):
                for k in range(3):
                    for c in range(2):
                        T=TW[lane,k,c]
                        if T<0: continue
                        for cc in range(2):
                            if TW[lane,k+1,cc]==T: W[lane,k+1,cc]+=W[lane,k,c]; TW[lane,k,c]=-1; W[lane,k,c]=0; break
            # cross half
            for lane in range(32,64):
                for c in range(2):
                    T=TW[lane-32,3,c]
                    if T<0: continue
                    done=False
                    for j in (0,1):
                        if done: break
                        for cc in range(2):
                            if TW[lane,j,cc]==T: W[lane,j,cc]+=W[lane-32,3,c]; TW[lane-32,3,c]=-1; W[lane-32,3,c]=0; done=True; break
        else:
            # current kernel: strict pattern
            on=TW[:,:,0]>=0
            FXl=TW[:,:,0]&0xffff; FYl=TW[:,:,0]>>16; CYl=TW[:,:,1]>>16; CXl=TE[:,:,0]&0xffff
            eg=np.zeros((64,4),bool); swg=np.zeros((64,4),bool)
            for lane in range(64):
                if lane&31==0: continue
                for k in range(4):
                    if on[lane,k] and on[lane-1,k] and CXl[lane-1,k]==FXl[lane,k] and FYl[lane-1,k]==FYl[lane,k] and CYl[lane-1,k]==CYl[lane,k]:
                        W[lane,k,0]+=E[lane-1,k,0]; W[lane,k,1]+=E[lane-1,k,1]; eg[lane-1,k]=True
            for lane in range(64):
                for k in range(3):
                    if on[lane,k] and on[lane,k+1] and FXl[lane,k]==FXl[lane,k+1] and CYl[lane,k]==FYl[lane,k+1]:
                        W[lane,k+1,0]+=W[lane,k,1]; swg[lane,k]=True
            for lane in range(32,64):
                if on[lane,0] and on[lane-32,3] and FXl[lane-32,3]==FXl[lane,0] and CYl[lane-32,3]==FYl[lane,0]:
                    W[lane,0,0]+=W[lane-32,3,1]; swg[lane-32,3]=True
            for lane in range(64):
                for k in range(4):
                    if eg[lane,k]: TE[lane,k]=-1; E[lane,k]=0
                    if swg[lane,k]: TW[lane,k,1]=-1; W[lane,k,1]=0
        for lane in range(64):
            for k in range(4):
                for c in range(2):
                    if TW[lane,k,c]>=0: pend.append((TW[lane,k,c],W[lane,k,c],'w' if (general or c==0) else 'a')); log.append((int(TW[lane,k,c]),wv,lane,k,c,'w'))
                    if TE[lane,k,c]>=0: pend.append((TE[lane,k,c],E[lane,k,c],'a')); log.append((int(TE[lane,k,c]),wv,lane,k,c,'e'))
    if census: return log
    # ownership rounds. mode 0: west corners claim, winners store, rest atomics (round 3 kernel before the east corners joined);
    # mode 1: every corner claims in round 1; mode 2: mode 1 + a second claim round whose winners add with plain read-modify-write
    for i,(T,v,kind) in enumerate(pend):
        final[T]=final.get(T,0)+v
    alive=list(range(len(pend)))
    cand=[i for i in alive if (pend[i][2]=='w' or MODE>=1)]
    for i in cand: owner[pend[i][0]]=i
    win=set(i for i in cand if owner[pend[i][0]]==i)
    tot_store+=len(win)
    rest=[i for i in alive if i not in win]
    if MODE>=2:
        owner2={}
        for i in rest: owner2[pend[i][0]]=i
        win2=set(i for i in rest if owner2[pend[i][0]]==i)
        tot_store+=0
        globals()['RMW']=globals().get('RMW',0)+len(win2)
        rest=[i for i in rest if i not in win2]
    tot_atomic+=len(rest)
    assert final.keys()==contrib.keys() and all(abs(final[t]-contrib[t])<1e-9 for t in contrib), "conservation"
    return tot_store,tot_atomic
TILES = ((0, 0), (320, 640), (256, 352), (480, 896), (672, 1248), (160, 96))


def report():
    """positional = the round-2 merge rule (same rows, same corner); texel = the texel-id rule of warp_splat_windows_kernel, with the ownership
    variants: 0 west corners claim a texel, the rest is added atomically (the kernel), 1 east corners claim too, 2 a second claim round whose
    winners add with a plain read-modify-write (neither built: the kernel is VALU-bound, not LDS-bound, after the merge)."""
    global MODE, RMW
    for tx in (0.1, 0.3):
        G = geom(tx)
        for general, modes in ((False, (0,)), (True, (0, 1, 2))):
            for MODE in modes:
                RMW = 0
                S = A = n = 0
                for (ty0, tx0) in TILES:
                    s_, a_ = sim_tile(*G, ty0, tx0, general)
                    S += s_; A += a_; n += 1024
                print(f"camera tx {tx}: {'texel-id' if general else 'positional'} merge, ownership mode {MODE}: stores/pixel {S / n:.3f}, "
                      f"rmw/pixel {RMW / n:.3f}, atomic corners/pixel {A / n:.3f}")


def conflict_census(tx=0.3, tile=(320, 640)):
    """Who still shares a texel after the register merges (why a corner is left for the atomics)."""
    import collections
    G = geom(tx)
    global MODE
    MODE = 0
    pend_log = []
    orig = sim_tile.__globals__.get('_census')
    FX, CX, FY, CY = G
    ty0, tx0 = tile
    # re-run the merge of sim_tile but keep (wave, lane, k, c, kind) per pending corner
    out = collections.Counter()
    per_tex = collections.defaultdict(list)
    for rec in sim_tile(FX, CX, FY, CY, ty0, tx0, True, census=True):
        per_tex[rec[0]].append(rec[1:])
    for T, lst in per_tex.items():
        if len(lst) < 2:
            continue
        lst.sort()
        for a, b in zip(lst, lst[1:]):
            (w1, l1, k1, c1, kd1), (w2, l2, k2, c2, kd2) = a, b
            if w1 != w2:
                out['different waves'] += 1
            elif l1 == l2:
                out[f'same thread rows {k1},{k2}'] += 1
            elif abs(l1 - l2) == 32:
                out[f'lane +-32 rows {k1},{k2}'] += 1
            elif abs(l1 - l2) == 1:
                out[f'neighbour lanes kinds {kd1}{kd2} rows {k1},{k2}'] += 1
            else:
                out[f'lanes {abs(l1-l2)} apart'] += 1
    return out


if __name__ == "__main__":
    report()
    print("texel conflicts left in a background tile:", dict(conflict_census()))
