"""Dev tool (GPU box): the arithmetic modes of the I3D plan side by side -- logits / probabilities / input gradient /
Grad-CAM against the exact-fp32 mode on one synthetic clip, then the time of a short search at batch B per mode.
usage: python tools/mode_compare.py [B] [iters] [frames] [modes,comma-separated]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import ivf_engine   # noqa: E402
import ivf_recipe as R   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["fp32", "bf16x6", "bf16x3", "bf16act"]
sd = R.i3d_state_dict(num_classes=174)
sml = "" if T == 16 else "none"
x1 = torch.from_numpy(R.clip(7, 3, T, 224, 224))[None].cuda()
ref = None
for m in modes:
    eng = ivf_engine.I3DEngine(174, (3, T, 224, 224), max_batch=B, softmax=True, math=m, stride_mod_layers=sml)
    t0 = time.perf_counter()
    eng.load_state_dict(sd, autotune=True)
    torch.cuda.synchronize()
    t_tune = time.perf_counter() - t0
    probs, logits = eng.forward(x1, want_logits=True)
    tgt = int(torch.argmax(probs[0]))
    score, dx = eng.backward(1, target=[tgt])
    cam, _ = eng.gradcam(x1, [tgt])
    out = dict(logits=logits.double().cpu().numpy(), probs=probs.double().cpu().numpy(), dx=dx.double().cpu().numpy(),
               cam=cam.double().cpu().numpy())
    if ref is None:
        ref = out
    rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
    l2 = float(np.linalg.norm(out["dx"] - ref["dx"]) / np.linalg.norm(ref["dx"]))
    ok = np.isfinite(ref["cam"]) & np.isfinite(out["cam"])
    xb = torch.stack([torch.from_numpy(R.clip(i % 16, 3, T, 224, 224)) for i in range(B)]).cuda()
    tg = eng.argmax(eng.forward(xb))
    raw = torch.zeros(B, T, device="cuda")
    eng.search(xb, tg, raw, 0.01, 0.02, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.search(xb, tg, raw, 0.01, 0.02, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{m:8s} T={T} logits {rel(out['logits'], ref['logits']):.2e} probs {rel(out['probs'], ref['probs']):.2e} "
          f"dx max {rel(out['dx'], ref['dx']):.2e} L2 {l2:.2e} cam max|d| {np.max(np.abs(out['cam'][ok] - ref['cam'][ok])):.2e} | "
          f"tune {t_tune:.1f}s, {dt * 1e3:.2f} ms/iteration at B={B} -> {B / (dt * 300 * (1 + 12 / 600)):.2f} clips/s est. (300 it)",
          flush=True)
    del eng
    torch.cuda.empty_cache()
