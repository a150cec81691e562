import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'pmx_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: take the last 8 pmx kernels
t0=None
for r in rows[-int(sys.argv[2]):]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print("%-40s grid %8s start %8.3f ms end %8.3f ms dur %7.3f" % (r['Kernel_Name'][5:45], r['Grid_Size_X'], (s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6))
