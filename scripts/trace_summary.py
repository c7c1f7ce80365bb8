import csv,glob,sys
d=sys.argv[1]
f=glob.glob(d+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'].replace('hipkkt::','').replace('void ','')[:26], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Start_Timestamp']), int(r['End_Timestamp']), r['VGPR_Count'], r['SGPR_Count'], r['Scratch_Size']) for r in rows]
which=sys.argv[2] if len(sys.argv)>2 else 'factor'
if which=='factor':
    idx=[i for i,s in enumerate(seq) if s[0].startswith(('k_front_wave','k_panel','k_schur','k_factor'))]
else:
    idx=[i for i,s in enumerate(seq) if s[0].startswith(('k_fwd','k_bwd'))]
# last contiguous run
end=idx[-1]; start=end
while start-1 in idx: start-=1
tot=0
prev=None
for s in seq[start:end+1]:
    gap=(s[3]-prev)/1000.0 if prev else 0
    print('%-28s %8.2f us  wgs %7d  vgpr %s sgpr %s scr %s gap %.1f'%(s[0],s[1],s[2],s[5],s[6],s[7],gap)); tot+=s[1]; prev=s[4]
print('sum of durations %.1f us; wall %.1f us'%(tot,(seq[end][4]-seq[start][3])/1000.0))
