"""Reads a rocprofv3 --kernel-trace CSV of `bench.py` and reports, per dense query launch: its duration, the gap to the previous one, and how much kernel time of
OTHER kernels ran inside the query's interval (i.e. beside it on another stream) vs inside the gap."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
q = [r for r in rows if 'avatar_kernel' in r['Kernel_Name']]
oth = [r for r in rows if 'avatar_kernel' not in r['Kernel_Name']]
prev_end = None
for i, r in enumerate(q):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    inside = sum(min(e, int(o['End_Timestamp'])) - max(s, int(o['Start_Timestamp'])) for o in oth if int(o['Start_Timestamp']) < e and int(o['End_Timestamp']) > s)
    gap = (s - prev_end) / 1e6 if prev_end else 0.0
    ingap = sum(int(o['End_Timestamp']) - int(o['Start_Timestamp']) for o in oth if prev_end and prev_end <= int(o['Start_Timestamp']) and int(o['End_Timestamp']) <= s) / 1e6
    qid = r.get('Queue_Id', '?')
    print(f'query {i}: {(e - s) / 1e6:7.2f} ms  gap before {gap:6.2f} ms (other kernels in the gap: {ingap:5.2f} ms)  other kernels beside it: {inside / 1e6:6.2f} ms  queue {qid}')
    prev_end = e
