import numpy as np
rng=np.random.default_rng(1)
n,m,dens=200,400,0.05
def model(nrows, ncols, mask):
    # mask[nrows, ncols]: rows are the "groups" (rows for A x, columns for A'w)
    lens=mask.sum(1)
    KC=[2,4,6,8,10,12,14]
    for K in KC:
        g=[max(1,int(2**np.ceil(np.log2(max(1,np.ceil(l/K)))))) for l in lens]
        if sum(g)<=512: break
    lanes=[]  # list of index arrays per lane
    for r in range(nrows):
        idx=np.nonzero(mask[r])[0]
        per=int(np.ceil(len(idx)/g[r])) if len(idx) else 0
        for p in range(g[r]):
            lanes.append(idx[p*per:(p+1)*per])
    while len(lanes)<512: lanes.append(np.array([],int))
    return K,lanes
def cycles(lanes,K,order):
    tot=0;ideal=0
    for w in range(8):
        L=lanes[64*w:64*w+64]
        ent=[order(l,i) for i,l in enumerate(L)]
        for k in range(K):
            for h in range(2):
                addrs=set()
                for l in ent[32*h:32*h+32]:
                    a=int(l[k]) if k<len(l) else 0
                    addrs.add(a)
                banks={}
                for a in addrs: banks[a%32]=banks.get(a%32,0)+1
                tot+=max(banks.values()); ideal+=1
    return tot,ideal
def plain(l,i): return l
def sorted_rot(l,i):
    if len(l)==0: return l
    s=sorted(l,key=lambda a:a%32)
    r=(i*len(s))//32 % len(s) if False else (i%len(s))
    return np.array(s[r:]+s[:r])
def by_bank_slot(l,i):
    # greedy: place entry in step whose target bank window matches: step k prefers bank (k*32//K + i) %32 ...
    return l
for name,(mask) in {"A x~ (rows gather x)":None,"A'w (cols gather w)":None}.items():
    pass
res={}
for trial in range(3):
    A=rng.random((m,n))<dens
    for nm,mask in (("Ax",A),("A'w",A.T)):
        K,lanes=model(mask.shape[0],mask.shape[1],mask)
        c0=cycles(lanes,K,plain); c1=cycles(lanes,K,sorted_rot)
        print(nm,"K",K,"plain",c0,"sorted+rot",c1, "avg len", np.mean([len(l) for l in lanes]))

def greedy_cycles(lanes,K,restrict=True):
    tot=0
    for w in range(8):
        for h in range(2):
            L=lanes[64*w+32*h:64*w+32*h+32]
            occ=[dict() for _ in range(K)]   # step -> bank -> set(addresses)
            place=[[None]*K for _ in L]
            # interleaved: round-robin over entries
            for e in range(K):
                for li,l in enumerate(L):
                    if e>=len(l): continue
                    a=int(l[e]); b=a%32
                    nsteps=len(l) if restrict else K
                    start=(e) % nsteps
                    done=False
                    for t in range(nsteps):
                        s=(start+t)%nsteps
                        if place[li][s] is not None: continue
                        if b in occ[s] and a not in occ[s][b]: continue
                        occ[s].setdefault(b,set()).add(a); place[li][s]=a; done=True; break
                    if not done:
                        # choose free step with the smallest load on bank b
                        best=None
                        for s in range(nsteps):
                            if place[li][s] is None:
                                ld=len(occ[s].get(b,()))
                                if best is None or ld<best[0]: best=(ld,s)
                        s=best[1]; occ[s].setdefault(b,set()).add(a); place[li][s]=a
            for s in range(K):
                # padding lanes read address 0
                banks={b:len(v) for b,v in occ[s].items()}
                if any(place[li][s] is None for li in range(len(L))):
                    if 0 not in occ[s].get(0,set()): banks[0]=banks.get(0,0)+1
                tot+=max(banks.values()) if banks else 1
    return tot
for trial in range(3):
    A=rng.random((m,n))<dens
    for nm,mask in (("Ax",A),("A'w",A.T)):
        K,lanes=model(mask.shape[0],mask.shape[1],mask)
        print(nm,"plain",cycles(lanes,K,plain)[0],"greedy restricted",greedy_cycles(lanes,K,True),"greedy any step",greedy_cycles(lanes,K,False),"ideal",2*8*K)

def rounds_algo(lanes, KR=14, ROUNDS=64):
    tot=0; maxr=0
    for w in range(8):
        for h in range(2):
            L=lanes[64*w+32*h:64*w+32*h+32]
            taken=[0]*32
            used=[0]*32; e=[0]*32
            place=[[None]*KR for _ in L]
            r=0
            while r<ROUNDS and any(e[i]<len(L[i]) for i in range(32)):
                r+=1
                props={}
                for i in range(32):
                    if e[i]>=len(L[i]): continue
                    a=int(L[i][e[i]]); b=a%32
                    avail=~(taken[b]|used[i]) & ((1<<KR)-1)
                    if avail:
                        start=(e[i]+i)%KR
                        hi=(avail>>start)<<start
                        x=hi if hi else avail
                        s=(x&-x).bit_length()-1
                        props.setdefault((s,b),[]).append(i)
                    else:
                        x=~used[i] & ((1<<KR)-1); s=(x&-x).bit_length()-1
                        used[i]|=1<<s; place[i][s]=a; e[i]+=1
                for (s,b),ls in props.items():
                    i=min(ls); a=int(L[i][e[i]])
                    taken[b]|=1<<s; used[i]|=1<<s; place[i][s]=a; e[i]+=1
            maxr=max(maxr,r)
            for i in range(32):
                while e[i]<len(L[i]):
                    x=~used[i] & ((1<<KR)-1); s=(x&-x).bit_length()-1
                    used[i]|=1<<s; place[i][s]=int(L[i][e[i]]); e[i]+=1
            for s in range(KR):
                banks={}
                for i in range(32):
                    a=place[i][s] if place[i][s] is not None else 0
                    banks.setdefault(a%32,set()).add(a)
                tot+=max(len(v) for v in banks.values())
    return tot,maxr
print("rounds algorithm (KR=14 steps): passes, max rounds; plain with 14 steps for comparison")
for trial in range(3):
    A=rng.random((m,n))<dens
    for nm,mask in (("Ax",A),("A'w",A.T)):
        K,lanes=model(mask.shape[0],mask.shape[1],mask)
        pl=cycles(lanes,14,plain)[0]
        print(nm,"plain14",pl,"rounds",rounds_algo(lanes),"floor",2*8*14)

def rounds_algo2(lanes, KR=14, ROUNDS=64, pad="zero", startf=lambda e,i,KR:(e+i)%KR, pick="first"):
    tot=0
    for w in range(8):
        for h in range(2):
            L=lanes[64*w+32*h:64*w+32*h+32]
            taken=[0]*32
            if pad=="zero_reserved":
                # bank 0 is pre-claimed in steps where some lane will pad?  unknown upfront; skip
                pass
            used=[0]*32; e=[0]*32
            place=[[None]*KR for _ in L]
            r=0
            while r<ROUNDS and any(e[i]<len(L[i]) for i in range(32)):
                r+=1
                props={}
                for i in range(32):
                    if e[i]>=len(L[i]): continue
                    a=int(L[i][e[i]]); b=a%32
                    avail=~(taken[b]|used[i]) & ((1<<KR)-1)
                    if avail:
                        start=startf(e[i],i,KR)
                        hi=(avail>>start)<<start
                        x=hi if hi else avail
                        s=(x&-x).bit_length()-1
                        props.setdefault((s,b),[]).append(i)
                    else:
                        x=~used[i] & ((1<<KR)-1); s=(x&-x).bit_length()-1
                        used[i]|=1<<s; place[i][s]=a; e[i]+=1
                for (s,b),ls in props.items():
                    i=min(ls); a=int(L[i][e[i]])
                    taken[b]|=1<<s; used[i]|=1<<s; place[i][s]=a; e[i]+=1
            for s in range(KR):
                banks={}
                for i in range(32):
                    if place[i][s] is None:
                        if pad=="zero": a=0
                        elif pad=="none": continue
                        elif pad=="own":
                            a=int(L[i][0]) if len(L[i]) else 0
                        elif pad=="lane": a=i   # element i: bank i
                    else: a=place[i][s]
                    banks.setdefault(a%32,set()).add(a)
                tot+=max([len(v) for v in banks.values()]+[1])
    return tot
for trial in range(2):
    A=rng.random((m,n))<dens
    for nm,mask in (("Ax",A),("A'w",A.T)):
        K,lanes=model(mask.shape[0],mask.shape[1],mask)
        print(nm,"zero",rounds_algo2(lanes),"none",rounds_algo2(lanes,pad="none"),"lane",rounds_algo2(lanes,pad="lane"),
              "start=e", rounds_algo2(lanes,pad="lane",startf=lambda e,i,KR:e%KR),
              "start=i", rounds_algo2(lanes,pad="lane",startf=lambda e,i,KR:i%KR))

def plain_pad(lanes,KR=14,pad="none"):
    tot=0
    for w in range(8):
        for h in range(2):
            L=lanes[64*w+32*h:64*w+32*h+32]
            for s in range(KR):
                banks={}
                for i in range(32):
                    if s<len(L[i]): a=int(L[i][s])
                    elif pad=="zero": a=0
                    else: continue
                    banks.setdefault(a%32,set()).add(a)
                tot+=max([len(v) for v in banks.values()]+[1])
    return tot
print("plain: zero-padding vs free padding; rounds+free padding")
for trial in range(3):
    A=rng.random((m,n))<dens
    for nm,mask in (("Ax",A),("A'w",A.T)):
        K,lanes=model(mask.shape[0],mask.shape[1],mask)
        print(nm,"plain zero",plain_pad(lanes,pad="zero"),"plain free",plain_pad(lanes),"rounds free",rounds_algo2(lanes,pad="none"))
