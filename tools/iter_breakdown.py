#!/usr/bin/env python3
"""Dev tool: kernel-time breakdown of ONE steady-state search iteration from a rocprofv3 kernel trace
(iterations are delimited by mask_reg_kernel launches; a middle one is taken)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows), key=lambda t: t[1])
starts = [i for i, k in enumerate(ks) if 'mask_reg_kernel' in k[0]]
mid = len(starts) * 3 // 4
it = ks[starts[mid]:starts[mid + 1]]
wall = it[-1][2] - it[0][1]
acc = collections.defaultdict(lambda: [0, 0])
busy = 0
for name, s, e in it:
    k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).replace("ivf::", "")
    acc[k][0] += e - s; acc[k][1] += 1; busy += e - s
print(f"iteration wall {wall/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms ({len(it)} launches, gaps {100*(wall-busy)/wall:.1f}%)")
fam = collections.Counter()
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    f = 'conv' if k.startswith('conv3d') else 'pool' if 'pool' in k else 'mask/other'
    fam[f] += t
    print(f"{t/1e3:9.1f} us {100*t/busy:5.1f}%  x{n:3d}  {k[:100]}")
print({k: f"{v/1e6:.2f} ms" for k, v in fam.items()})
