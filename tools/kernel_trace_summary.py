import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('ivf::', '')
    key = (k, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    acc[key][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); acc[key][1] += 1
for (k, g), (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{t/n/1e3:9.1f} us x{n:6d} grid {g:>9s}  {k}")
