import csv,glob,collections,sys
for d in sys.argv[1:]:
    f=glob.glob(d+'/runc/*_counter_collection.csv')[0]
    agg=collections.defaultdict(lambda:collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        agg[k]['_dur']+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))
        cnt[(k,r['Counter_Name'])]+=1
    names=sorted({c for k in agg for c in agg[k] if c!='_dur'})
    print(d); print('%-24s'%'kernel'+''.join('%22s'%n[-21:] for n in names))
    for k in agg:
        if not k.startswith('k_'): continue
        n=max(1,cnt[(k,names[0])])
        print('%-24s'%k[:24]+''.join('%22.0f'%(agg[k][c]/n) for c in names))
