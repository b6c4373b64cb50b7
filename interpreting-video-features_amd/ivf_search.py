"""Batched per-clip mask search on top of a network engine (I3D or CLSTM_4).

One call runs, for b DIFFERENT clips with b DIFFERENT masks, what the reference
runs for one `batch_index` at a time (FindMasksComparison_I3D_smth.py:166-277):
baseline scores -> init_mask('central') -> N Adam iterations -> reverse score ->
Grad-CAM.  In eval mode the rows of a batch are independent, so
per clip the numbers are the ones the reference computes.

Everything that computes is a libivf_hip call; this file only sequences them
and makes the handful of host decisions the reference makes on the host
(which central mask to keep, mask.py:134-147).
"""
import numpy as np
import torch

import ivf_lib as L


def central_masks(T, device):
    """mask.py:135-137 for i = 1 .. T//2-1: ones with i zeros at each end."""
    rows = []
    for i in range(1, T // 2):
        m = torch.ones(T)
        m[:i] = 0
        m[T - i:] = 0
        rows.append(m)
    return torch.stack(rows).to(device)


def init_masks_central(engine, x, target, orig_score, threshold=0.9, mask_type="freeze"):
    """mask.init_mask(mode='central'), mask.py:121-154, for b clips at once.
    Returns (raw masks [b,T] in {-5,+5}, info dict).  The reference stops at the
    first i whose score ratio drops below `threshold`; here all T//2-1 candidates are
    scored (<= 6 extra forwards per clip) and the same i is selected afterwards, so
    the loop needs one host sync instead of one per candidate."""
    b, _, T = x.shape[0], x.shape[1], x.shape[2]
    dev = x.device
    idx = torch.arange(b, device=dev)
    tl = target.long()
    # the fully perturbed clip is ALWAYS the fully frozen one (mask.py:123-128), whatever mask_type
    full = engine.perturbed_forward(x, torch.ones(b, T, device=dev), "freeze")[idx, tl]
    cands = central_masks(T, dev)
    cen = torch.stack([engine.perturbed_forward(x, cands[i][None].expand(b, T).contiguous(), mask_type)[idx, tl]
                       for i in range(cands.shape[0])], dim=1)                                # :139-141
    n = cands.shape[0]
    orig = L.f32c(orig_score)
    full, cen = L.f32c(full), L.f32c(cen)
    raw = torch.empty(b, T, device=dev)
    first = torch.empty(b, dtype=torch.int32, device=dev)
    ratio = torch.empty(b, n, device=dev)
    with torch.cuda.device(dev):                                                               # :142-154
        L.check(L.lib().ivf_init_central_select(L.ptr(orig), L.ptr(full), L.ptr(cen), b, n, T, float(threshold),
                                                L.ptr(raw), L.ptr(first), L.ptr(ratio), L.stream()))
    return raw, dict(full=full, central=cen, ratio=ratio, chosen_i=first.long())


def find_submasks_host(mask_row, thresh=0.1):
    """Host view of the integer output of ivf_submask_pairs for one mask."""
    T = mask_row.numel()
    m = L.f32c(mask_row.detach().reshape(-1))
    L.require_gpu(m)
    run = torch.empty(T, dtype=torch.int32, device=m.device)
    partner = torch.empty(T, dtype=torch.int32, device=m.device)
    weight = torch.empty(T, device=m.device)
    with torch.cuda.device(m.device):
        L.check(L.lib().ivf_submask_pairs(L.ptr(m), T, float(thresh), L.ptr(run), L.ptr(partner), L.ptr(weight),
                                          L.stream()))
    runs = {}
    for t, r in enumerate(run.cpu().tolist()):
        if r >= 0:
            runs.setdefault(r, []).append(t)
    return [runs[r] for r in sorted(runs)]


def frame_ranking(mask):
    """Integer frame-importance ranking: stable argsort of -mask (ivf_rank_frames)."""
    m = L.f32c(mask.detach())
    L.require_gpu(m)
    T = m.shape[-1]
    order = torch.empty(m.shape, dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        L.check(L.lib().ivf_rank_frames(L.ptr(m), m.numel() // T, T, L.ptr(order), L.stream()))
    return order.long()


class MaskSearch:
    def __init__(self, engine, lam1=0.01, lam2=0.02, n_iter=300, mask_type="freeze", threshold=0.9,
                 lr=0.2, grad_cam_type="guessed", do_gradcam=True, run_temp_mask=True,
                 normalize_per_frame=True, gradcam_size=None):
        self.engine = engine
        self.lam1, self.lam2, self.n_iter = float(lam1), float(lam2), int(n_iter)
        self.mask_type, self.threshold, self.lr = mask_type, threshold, lr
        self.grad_cam_type = grad_cam_type
        self.do_gradcam, self.run_temp_mask = do_gradcam, run_temp_mask
        self.normalize_per_frame = normalize_per_frame
        self.gradcam_size = gradcam_size

    def run(self, x, labels, want_traj=False):
        """x [b,C,T,H,W] float32 on the GPU, labels [b] ints.  Returns a dict of device
        tensors, one row per clip (nothing is copied to the host here)."""
        eng = self.engine
        b = x.shape[0]
        dev = x.device
        labels = torch.as_tensor(labels, device=dev).to(torch.int32).reshape(-1)
        out = {}
        probs = eng.forward(x)                                               # smth:176
        pred = eng.argmax(probs)                                             # smth:181 / :217
        target = pred if self.grad_cam_type == "guessed" else labels         # smth:179-184
        idx = torch.arange(b, device=dev)
        out["pred_class"] = pred
        out["target"] = target
        out["original_score_guess"] = probs[idx, pred.long()]
        out["original_score_true"] = probs[idx, labels.long()]
        if self.run_temp_mask:
            raw, info = init_masks_central(eng, x, target, probs[idx, target.long()], self.threshold,
                                           self.mask_type)                   # smth:188-190
            out["init_mask"] = raw.clone()
            traj, _ = eng.search(x, target, raw, self.lam1, self.lam2, self.n_iter, lr=self.lr,
                                 want_traj=True, mode=self.mask_type)        # smth:191-214
            mask = torch.empty_like(raw)                                     # smth:216, with the sigmoid the loop itself uses
            with torch.cuda.device(dev):
                L.check(L.lib().ivf_sigmoid(L.ptr(raw), L.ptr(mask), raw.numel(), L.stream()))
            out["time_mask"] = mask
            out["freeze_score"] = traj[-1, :, 3] if self.n_iter > 0 else torch.full((b,), float("nan"), device=dev)
            rev = eng.perturbed_forward(x, mask, "reverse")                  # smth:234-235
            out["reverse_score"] = rev[idx, target.long()]
            out["ranking"] = frame_ranking(mask)
            out["snapped"] = mask > 0.5                                      # mask.py:5-10
            if want_traj:
                out["traj"] = traj
        if self.do_gradcam:
            gc_target = pred if self.grad_cam_type == "guessed" else target  # smth:253,266-267
            cam, _ = eng.gradcam(x, gc_target, per_frame=self.normalize_per_frame, out_hw=self.gradcam_size)
            out["gradcam"] = cam                                             # smth:269
        return out


RECORD_INT_FIELDS = ("clip_id", "pred_class", "target")
RECORD_FLOAT_FIELDS = ("original_score_guess", "original_score_true", "freeze_score", "reverse_score")
RECORD_FIELDS = RECORD_INT_FIELDS + RECORD_FLOAT_FIELDS


def pack_records(clip_ids, res, T, with_cam=False):
    """Fixed-size per-clip record [b, 7+T (+ cam)] int32 for the all-gather: clip id,
    pred_class, target as integers; the four scores and sigma(mask)[T] as bit-cast float32.
    with_cam=True appends the clip's Grad-CAM map res["gradcam"] [T', h, w] (bit-cast float32, row-major) -- the
    optional payload of SURVEY.md 8e; at [16,224,224] that is 3.2 MB per clip, a few milliseconds of xGMI time per
    step of 32 clips per GPU against seconds of search.  Every rank must use the same map shape."""
    dev = res["pred_class"].device
    ints = [torch.as_tensor(clip_ids, device=dev).to(torch.int32)]
    ints += [res[k].to(torch.int32) for k in RECORD_INT_FIELDS[1:]]
    flt = torch.stack([res[k].float() for k in RECORD_FLOAT_FIELDS], dim=1)
    parts = [flt, res["time_mask"].float()]
    if with_cam:
        cam = res["gradcam"]
        parts.append(cam.float().reshape(cam.shape[0], -1))
    flt = torch.cat(parts, dim=1).contiguous()
    return torch.cat([torch.stack(ints, dim=1), flt.view(torch.int32)], dim=1).contiguous()


def unpack_record(row, T, cam_shape=None):
    """One gathered row -> dict; cam_shape = (T', h, w) when the rows carry Grad-CAM maps (pack_records(with_cam=True))."""
    row = row.detach().cpu().contiguous()
    ni = len(RECORD_INT_FIELDS)
    d = {k: int(row[i]) for i, k in enumerate(RECORD_INT_FIELDS)}
    f = row[ni:].view(torch.float32)
    for i, k in enumerate(RECORD_FLOAT_FIELDS):
        d[k] = float(f[i])
    nf = len(RECORD_FLOAT_FIELDS)
    d["time_mask"] = f[nf:nf + T].numpy().copy()
    if cam_shape is not None:
        n = int(np.prod(cam_shape))
        if f.numel() != nf + T + n:
            raise L.IvfError(f"record has {f.numel() - nf - T} map values, cam_shape {tuple(cam_shape)} needs {n}")
        d["gradcam"] = f[nf + T:].numpy().reshape(cam_shape).copy()
    return d
