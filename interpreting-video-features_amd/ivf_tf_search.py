"""Per-clip flow of the reference's TF-half driver (video_features_tf/mask/find_mask_kth.py:380-470) on a
`TFCLSTMEngine`: SURVEY 8f N4, a documented extension whose parity with TensorFlow cannot be pinned (see
csrc/tf_clstm.hip).  Everything that computes is a libivf_hip call.

Quirks of the TF graph that are reproduced on purpose: the mask variable ALWAYS passes through a sigmoid before the
freeze recurrence, so the "unperturbed" baseline prediction (mask_var = 0) is the clip frozen with 0.5, the "fully
perturbed" score of init_mask uses sigmoid(1) = 0.73, and Grad-CAM (gradcam.py:51-56) runs on the 0.5-frozen clip.
"""
import torch

import ivf_lib as L


def _sig(v):
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        L.check(L.lib().ivf_sigmoid(L.ptr(v), L.ptr(out), v.numel(), L.stream()))
    return out


def init_mask_central(engine, x, target, thresh=0.9):
    """mask.py:83-130 (TF half) for b clips: scores through sigmoid(mask_var); returns raw masks in {-5, +5}."""
    b, T = x.shape[0], x.shape[2]
    dev = x.device
    idx, tl = torch.arange(b, device=dev), target.long()

    def score(mv):
        return engine.perturbed_forward(x, _sig(mv.expand(b, T).contiguous()))[idx, tl]
    full, orig = score(torch.ones(1, T, device=dev)), score(torch.zeros(1, T, device=dev))
    cands = []
    for i in range(1, T // 2):
        m = torch.ones(T, device=dev)
        m[:i] = 0
        m[T - i:] = 0
        cands.append(m)
    cands = torch.stack(cands)
    cen = torch.stack([score(c[None]) for c in cands], dim=1)
    below = (orig[:, None] - cen) / (orig[:, None] - full[:, None]) < thresh
    first = torch.where(below.any(dim=1), below.float().argmax(dim=1), torch.full((b,), cands.shape[0] - 1, device=dev))
    chosen = cands[first]
    return torch.where(chosen == 0, torch.tensor(-5.0, device=dev), torch.tensor(5.0, device=dev)).contiguous()


def find_mask(engine, x, labels, lam1=0.01, lam2=0.02, n_iter=100, lr=0.2, focus_type="correct",
              normalization_mode="frame", do_gradcam=True, min_score=0.1):
    """find_mask_kth.py:395-470 for b clips at once.  labels [b] ints.  Returns a dict of device tensors:
    `skipped` marks clips whose true-class score is below 0.1 (the driver `continue`s on them, :411-413)."""
    b, T = x.shape[0], x.shape[2]
    dev = x.device
    labels = torch.as_tensor(labels, device=dev).to(torch.int32).reshape(-1)
    half = torch.full((b, T), 0.5, device=dev)                       # sigmoid(mask_var = 0)
    preds = engine.perturbed_forward(x, half)                        # :401-404 (softmax: `output`, :419-423)
    idx = torch.arange(b, device=dev)
    guessed = preds.argmax(dim=1).to(torch.int32)
    target = labels if focus_type == "correct" else guessed          # :361-364
    out = {"output": preds, "pred_class": guessed, "target": target,
           "skipped": preds[idx, labels.long()] < min_score}
    raw = init_mask_central(engine, x, labels)                       # :426-431 (init_mask looks at argmax(label))
    out["init_mask"] = raw.clone()
    traj, _ = engine.search(x, target, raw, lam1, lam2, n_iter, lr=lr)
    out["traj"] = traj
    out["time_mask"] = _sig(raw)                                     # :457
    if do_gradcam:
        cam, _ = engine.gradcam(x, target, mask=half, normalization_mode=normalization_mode)
        out["gradcam"] = cam
    return out
