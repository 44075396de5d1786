import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'adam_multi_kernel' in r['Kernel_Name']]
tot={}
for k in range(3,9):
    a,b=marks[-k-1]+1,marks[-k]+1
    step=rows[a:b]
    qs={}
    for r in step: qs.setdefault(r['Queue_Id'],[]).append(r)
    q1=max(qs.values(),key=len)
    ig=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in q1 if 'mlp_fwd_kernel<128, 32, true, true, 0, 2>' in r['Kernel_Name']]
    fin=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in q1 if 'bn_rows_bwd_kernel<false>' in r['Kernel_Name']]
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in q1)/1e6
    print('step -%d: q1 busy %.3f ms; IG sum %.0f us first4 %s ; finish sum %.0f first3 %s'%(k,busy,sum(ig),[round(x) for x in ig[:4]],sum(fin),[round(x) for x in fin[:3]]))
