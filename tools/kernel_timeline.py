"""Timeline of the last 40 kernels longer than 0.3 ms of a rocprofv3 kernel trace (queue, start, duration).  Usage: kernel_timeline.py <prefix>_kernel_trace.csv"""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=[r for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>300000]
t0=int(big[-min(40,len(big))]['Start_Timestamp'])
for r in big[-40:]:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:28]
    print("%-28s q%-3s start %8.2f dur %7.2f ms"%(n, r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
