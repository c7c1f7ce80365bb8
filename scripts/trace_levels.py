import csv,glob,sys,collections
d=sys.argv[1]; which=sys.argv[2] if len(sys.argv)>2 else 'factor'
f=glob.glob(d+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'].replace('hipkkt::','').replace('void ','').split('(')[0], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])) for r in rows]
names=('k_front_wave','k_panel','k_schur') if which=='factor' else ('k_fwd','k_bwd')
idx=[i for i,s in enumerate(seq) if s[0].startswith(names)]
end=idx[-1]; start=end
while start-1 in idx: start-=1
tot=collections.defaultdict(float)
line=[]
for s in seq[start:end+1]:
    tot[s[0]]+=s[1]
    line.append('%s %.0fus/%d'%(s[0].replace('k_',''),s[1],s[2]))
print('  '.join(line))
print({k:round(v,1) for k,v in tot.items()}, 'total', round(sum(tot.values()),1))
