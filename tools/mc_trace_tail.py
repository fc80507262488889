import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
mc=[r for r in rows if 'mc_' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
for r in mc[-12:]:
    print(r['Kernel_Name'].split('(')[0][-40:].ljust(40), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
