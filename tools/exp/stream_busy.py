import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 10 lift kernels define steady-state window
idx=[i for i,r in enumerate(rows) if 'lift_knn_gather' in r['Kernel_Name']]
a,b=idx[-11],idx[-1]
t0,t1=int(rows[a]['Start_Timestamp']),int(rows[b]['Start_Timestamp'])
span=(t1-t0)/10/1e3
busy=collections.defaultdict(float)
names=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows[a:b]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    q=r['Queue_Id']+'/'+r.get('Stream_Id','')
    busy[q]+=d/10
    n=r['Kernel_Name']; n=n[n.find('::')+2:] if n.startswith('void (anon') else n
    names[q][n[:60]]+=d/10
print('span per batch %.1f us'%span)
for q,v in busy.items():
    print('queue',q,'busy %.1f us/batch'%v)
    for n,d in sorted(names[q].items(), key=lambda kv:-kv[1])[:8]:
        print('     %8.1f  %s'%(d,n))
