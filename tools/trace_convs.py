#!/usr/bin/env python3
"""Per-layer view of one search iteration from a rocprofv3 kernel trace (dev tool)."""
import csv, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'interpreting-video-features_amd'))
import ivf_arch as arch
path, B = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
ks = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X'])) for r in rows), key=lambda t: t[1])
starts = [i for i, k in enumerate(ks) if 'mask_reg_kernel' in k[0]]
it = ks[starts[len(starts)//2]:starts[len(starts)//2 + 1]]
ops = []
def conv(name, cin, cout, k, st, ss, T, H, W):
    To, Ho, Wo = arch.out_size(T, k, st), arch.out_size(H, k, ss), arch.out_size(W, k, ss)
    ops.append((name, 2.0 * To * Ho * Wo * cout * k**3 * cin, To * Ho * Wo, cout, cin * k**3)); return To, Ho, Wo
def pool(name, kt, ks_, st, ss, T, H, W):
    ops.append((name, 0, 0, 0, 0)); return arch.out_size(T, kt, st), arch.out_size(H, ks_, ss), arch.out_size(W, ks_, ss)
T, H, W = 16, 224, 224
T, H, W = conv('Conv3d_1a_7x7', 3, 64, 7, 2, 2, T, H, W); T, H, W = pool('MaxPool3d_2a', 1, 3, 1, 2, T, H, W)
T, H, W = conv('Conv3d_2b_1x1', 64, 64, 1, 1, 1, T, H, W); T, H, W = conv('Conv3d_2c_3x3', 64, 192, 3, 1, 1, T, H, W)
T, H, W = pool('MaxPool3d_3a', 1, 3, 1, 2, T, H, W)
for n in arch.ENDPOINTS[5:]:
    if n in arch.POOLS:
        k, s = arch.POOLS[n]; T, H, W = pool(n, k[0], k[1], s[0], s[1], T, H, W); continue
    cin, oc = arch.INCEPTION[n]
    # forward: ONE GEMM for b0 | b1a | b2a; backward: b0's op carries the fused backward GEMM, the group's is skipped
    conv(n+'.b0', cin, oc[0], 1, 1, 1, T, H, W); ops[-1] = ops[-1] + ('fusedbwd', 2.0 * T * H * W * cin * (oc[0] + oc[1] + oc[3]), 'skipfwd')
    conv(n+'.b0|1a|2a', cin, oc[0] + oc[1] + oc[3], 1, 1, 1, T, H, W); ops[-1] = ops[-1] + ('skipbwd', 0)
    conv(n+'.b1b', oc[1], oc[2], 3, 1, 1, T, H, W)
    conv(n+'.b2b', oc[3], oc[4], 3, 1, 1, T, H, W); pool(n+'.b3a', 3, 3, 1, 1, T, H, W); conv(n+'.b3b', cin, oc[5], 1, 1, 1, T, H, W)
seq = [('fwd',) + o[:5] for o in ops if not (len(o) > 7 and o[7] == 'skipfwd')]
for o in reversed(ops):
    if len(o) > 5 and o[5] == 'skipbwd': continue
    if len(o) > 5 and o[5] == 'fusedbwd': seq.append(('bwd', o[0] + '+1a+2a', o[6]) + o[2:5])
    else: seq.append(('bwd',) + o[:5])
kern = [k for k in it if 'conv3d_' in k[0] or 'maxpool' in k[0]]
assert len(seq) == len(kern), (len(seq), len(kern))
tot = totf = pool_t = 0; out = []
for s, k in zip(seq, kern):
    dur = (k[2] - k[1]) / 1e3; fl = s[2] * B
    var = (('H' if 'halo' in k[0] else '') + k[0].split('<')[1].split('>')[0].replace(' ', '')) if '<' in k[0] else ('pix4' if 'pix4' in k[0] else 'pool')
    out.append((dur, s[0], s[1], var, k[3] // 256, fl / dur / 1e6 if fl else 0, s[3] * B, s[4], s[5]))
    if fl: tot += dur; totf += fl
    else: pool_t += dur
print("conv us %.0f  TF %.1f   pools us %.0f   iteration us %.0f" % (tot, totf / tot / 1e6, pool_t, (it[-1][2] - it[0][1]) / 1e3))
cum = 0
for o in sorted(out, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    cum += o[0]
    print("%8.1f us %s %-20s %-22s wgs=%-6d %6.1f TF  M=%d Cout=%d K=%d  cum %.0f" % (o + (cum,)))
