import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id','')) for r in rows), key=lambda t: t[1])
starts = [i for i, k in enumerate(ks) if 'mask_reg_kernel' in k[0]]
mid = len(starts) * 3 // 4
it = ks[starts[mid]:starts[mid + 1]]
t0 = it[0][1]
for name, s, e, q in it[:60]:
    k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).replace("ivf::", "")
    print(f"{(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f}  ({(e-s)/1e3:7.1f} us) q{q[-3:]}  {k[:70]}")
print('wall', (it[-1][2]-t0)/1e6, 'ms; sum', sum(e-s for _,s,e,_ in it)/1e6)
