import re,bisect,sys
from collections import Counter
import numpy as np
for f in sys.argv[1:]:
    sn=[];leaves=[]
    for line in open(f):
        if line.startswith('[top]'):
            m=re.match(r"\[top\] level\s+(\d+) sn\s+(\d+) cols\s+(\d+)\.\.\+\s*(\d+) rows\s+(\d+)",line)
            l,s,c0,nc,rows=map(int,m.groups()); sn.append((c0,nc,l,rows))
        elif line.startswith('[ndleaf]'):
            a=line.split(); leaves.append((int(a[2]),int(a[4])))
    sn.sort(); starts=[x[0] for x in sn]; lv=np.array([x[2] for x in sn])
    H=Counter(); tall=[]
    for pos,sz in leaves:
        if sz<150: continue
        i=bisect.bisect_left(starts,pos); j=bisect.bisect_left(starts,pos+sz)
        l=int(lv[i:j].max()); H[l]+=1
        if l>=8: tall.append((pos,sz,l))
    print(f, 'nlevels', lv.max()+1, sorted(H.items())); print(tall[:12])
