import itertools, random
RG=[[*range(0,4),*range(12,16),*range(20,28)],[*range(4,12),*range(16,20),*range(28,32)],
    [*range(32,36),*range(44,48),*range(52,60)],[*range(36,44),*range(48,52),*range(60,64)]]
def read_ok(f):
    worst=1
    for t in range(4):
        for grp in RG:
            slots={}
            for l in grp:
                r=l&15; g=l>>4
                s=((g*4+t)^f[r])&15
                slots[s]=slots.get(s,0)+1
            worst=max(worst,max(slots.values()))
    return worst
def write_ok(f):
    worst=1
    for c in range(16):
        for base in (0,8):
            slots={}
            for r in range(base,base+8):
                s=(c^f[r])&7
                slots[s]=slots.get(s,0)+1
            worst=max(worst,max(slots.values()))
    return worst
f0=list(range(16)); print('identity',read_ok(f0),write_ok(f0))
f1=[r^(4 if 4<=r<=11 else 0) for r in range(16)]; print('f1',read_ok(f1),write_ok(f1))
# search over permutations of 16 by random + constraint
best=None
random.seed(1)
# structured: f(r) = M r (GF2 linear 4x4) ^ const
import numpy as np
cnt=0
for cols in itertools.product(range(16),repeat=4):
    f=[0]*16
    for r in range(16):
        v=0
        for b in range(4):
            if r>>b&1: v^=cols[b]
        f[r]=v
    if len(set(f))<16: continue
    a,b=read_ok(f),write_ok(f)
    if a==1 and b==1:
        print('linear',cols,f); cnt+=1
        if cnt>5: break
print('found',cnt)
