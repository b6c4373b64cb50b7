#!/usr/bin/env python3
"""Dev tool: load the I3D plan at batch B with IVF_TUNE_LOG set and summarise, per layer and
direction, the best variant of each kernel family (igemm / halo / pix4)."""
import collections
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
log = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tune.log"
if os.path.exists(log):
    os.remove(log)
os.environ["IVF_TUNE_LOG"] = log
import ivf_engine, ivf_recipe as R   # noqa: E402
math = sys.argv[3] if len(sys.argv) > 3 else None       # arithmetic mode (default: ivf_engine.DEFAULT_MATH)
eng = ivf_engine.I3DEngine(174, (3, 16, 224, 224), max_batch=B, softmax=True, math=math)
eng.load_state_dict(R.i3d_state_dict(num_classes=174))
fam = lambda v: "pix4" if v == 15 else "halo" if v >= 16 else "igemm"
best = collections.OrderedDict()
for ln in open(log):
    m = re.match(r"(\S+) (\S+) b=(\d+) variant=(\d+) ms=([\d.]+) gflop=([\d.]+)", ln)
    name, d, _, v, ms, gf = m.groups()
    e = best.setdefault((name, d), {"gflop": float(gf)})
    f = fam(int(v))
    if f not in e or float(ms) < e[f][0]:
        e[f] = (float(ms), int(v))
tot = collections.Counter()
for (name, d), e in best.items():
    fams = {f: e[f] for f in ("igemm", "halo", "pix4") if f in e}
    w = min(fams, key=lambda f: fams[f][0])
    tot["best"] += fams[w][0]
    if True:
        print(f"{name:22s} {d}  " + "  ".join(f"{f}:{fams[f][0]*1e3:7.0f}us(v{fams[f][1]})" for f in fams)
              + f"   best {w} {e['gflop']/fams[w][0]:.0f} TF")
print(f"sum of per-layer bests: {tot['best']:.2f} ms (B={B})")
