import re,sys
nodes={}
for line in open(sys.argv[1]):
    m=re.match(r"\[top\] level\s+(\d+) sn\s+(\d+) cols\s+(\d+)\.\.\+\s*(\d+) rows\s+(\d+) parent\s+(-?\d+) kids:(.*)",line)
    l,sn,c0,nc,rows,par=map(int,m.groups()[:6])
    kids=[(int(a),int(b),int(c)) for a,b,c in re.findall(r"(\d+)\(l(\d+),nc(\d+)\)",m.group(7))]
    nodes[sn]=(l,c0,nc,rows,par,kids)
sn=max(nodes,key=lambda s:nodes[s][0])
while sn in nodes:
    l,c0,nc,rows,par,kids=nodes[sn]
    big=[k for k in kids if k[1]>0]
    print("level",l,"sn",sn,"col0",c0,"nc",nc,"rows",rows,"nkids",len(kids),"big kids",[(k[0],k[1],k[2]) for k in big][:6])
    if not big: break
    sn=max(big,key=lambda k:k[1])[0]
