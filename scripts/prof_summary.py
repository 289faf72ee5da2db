import csv,glob,re,sys
d=sys.argv[1]; steps=float(sys.argv[2]) if len(sys.argv)>2 else 4
f=(glob.glob(d+'/*/*kernel_stats.csv')+glob.glob(d+'/*kernel_stats.csv'))[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms/step %.2f'%(tot/steps/1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    n=re.sub(r'_ZN12_GLOBAL__N_1\d+','',r['Name']); n=re.sub(r'\(anonymous namespace\)::','',n)[:52]
    print('%-52s n%4s ms/step %.3f avg_us %.1f'%(n,r['Calls'],float(r['TotalDurationNs'])/steps/1e6,float(r['AverageNs'])/1e3))
