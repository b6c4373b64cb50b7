"""Oracle (test infrastructure): temporal perturbation, TV/L1 regulariser, mask
initialisation, Adam and the per-clip search loop, restated on torch CPU fp32.

Follows video_features_pytorch/mask.py and
video_features_pytorch/FindMasksComparison_I3D_smth.py:176-251.
"""
import math

import torch


def freeze(seq, mask):
    """mask.py:11-22.  seq [B,C,T,H,W], mask [T] or [B,T] (per-clip masks, the
    batched generalisation of SURVEY.md F10).  P[0]=X[0];
    P[u]=(1-m[u])*X[u]+m[u]*P[u-1]; m[0] is unused."""
    T = seq.shape[2]
    m = mask if mask.dim() == 2 else mask[None, :].expand(seq.shape[0], T)
    frames = [seq[:, :, 0]]
    for u in range(1, T):
        mu = m[:, u].view(-1, 1, 1, 1)
        frames.append((1 - mu) * seq[:, :, u] + mu * frames[-1])
    return torch.stack(frames, dim=2)


def find_submasks_from_mask(mask, thresh=0.1):
    """mask.py:60-85: runs of consecutive frames with mask > thresh (strict)."""
    # compare in the mask's own dtype: `mask[j] > thresh` on a float32 tensor rounds the
    # Python scalar to float32 first (0.1f > 0.1 is False)
    on = (torch.as_tensor(mask) > thresh).tolist()
    runs, cur = [], None
    for j, v in enumerate(on):
        if v:
            if cur is None:
                cur = []
            cur.append(j)
        elif cur is not None:
            runs.append(cur)
            cur = None
    if cur is not None:
        runs.append(cur)
    return runs


def reverse(seq, mask, thresh=0.1):
    """mask.py:24-56.  Within each run, frame a=run[u] and b=run[-(u+1)] are
    blended with each other using m[a] for BOTH (mask.py:50-56); the middle
    frame of an odd run and all frames outside runs are copied."""
    out = seq.clone()
    for run in find_submasks_from_mask(mask, thresh):
        for u in range(len(run) // 2):
            a, b = run[u], run[-(u + 1)]
            ma = mask[a]
            out[:, :, a] = (1 - ma) * seq[:, :, a] + ma * seq[:, :, b]
            out[:, :, b] = (1 - ma) * seq[:, :, b] + ma * seq[:, :, a]
    return out


def snap(mask):
    """mask.py:5-10, in place on the caller's tensor."""
    with torch.no_grad():
        mask.copy_((mask > 0.5).to(mask.dtype))
    return mask


def perturb_sequence(seq, mask, perturbation_type='freeze', snap_values=False):
    """mask.py:4-57."""
    if snap_values:
        snap(mask)
    if perturbation_type == 'freeze':
        return freeze(seq, mask)
    if perturbation_type == 'reverse':
        return reverse(seq, mask)
    raise UnboundLocalError("perturbed_input")  # mask.py:57 returns an unset local


def calc_tv_norm(mask, p=3, q=3):
    """mask.py:88-100: edge pairs once, interior pairs twice; (val^(1/p))^q."""
    val = mask.new_zeros(())
    for u in range(1, len(mask) - 1):
        val = val + torch.abs(mask[u - 1] - mask[u]) ** p
        val = val + torch.abs(mask[u + 1] - mask[u]) ** p
    val = val ** (1 / p)
    return val ** q


def central_mask(T, i):
    """mask.py:135-137: ones with i zeros at each end (raw 0/1, no sigmoid)."""
    m = torch.ones(T)
    m[:i] = 0
    m[T - i:] = 0
    return m


def init_mask_central(seq, score_fn, threshold=0.9, mask_type='freeze'):
    """mask.py:121-154, device agnostic.  `score_fn(x)` returns the scalar the
    reference reads as model(x)[batch_index, target[batch_index]].
    Returns (mask_pm5 [T], info dict with the per-i scores)."""
    T = seq.shape[2]
    frozen = seq[:, :, :1].expand_as(seq).contiguous()       # :123-126
    full = float(score_fn(frozen))                           # :128
    orig = float(score_fn(seq))                              # :129
    scores, ratios = [], []
    new_mask = torch.ones(T)
    for i in range(1, T // 2):                               # :134
        new_mask = central_mask(T, i)
        cen = float(score_fn(perturb_sequence(seq, new_mask, mask_type)))
        scores.append(cen)
        # fp32 arithmetic as in the reference (tensors of dtype float32)
        ratio = float((torch.tensor(orig) - torch.tensor(cen)) / (torch.tensor(orig) - torch.tensor(full)))
        ratios.append(ratio)
        if ratio < threshold:                                # :143 (NaN compares False)
            break
    m = torch.where(new_mask == 0, torch.tensor(-5.0), torch.tensor(5.0))  # :149-154
    return m, dict(full=full, orig=orig, central=scores, ratios=ratios)


class Adam:
    """torch.optim.Adam defaults as used at FindMasksComparison_I3D_smth.py:191
    (lr 0.2, betas (0.9, 0.999), eps 1e-8, no weight decay), restated so the
    HIP kernel has an explicit formula to match (torch/optim/adam.py single-tensor
    path: denom = sqrt(v)/sqrt(1-b2^t) + eps; step = lr/(1-b1^t))."""

    def __init__(self, param, lr=0.2, b1=0.9, b2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = param, lr, b1, b2, eps
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.t = 0

    def step(self, grad):
        self.t += 1
        self.m.mul_(self.b1).add_(grad, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(grad, grad, value=1 - self.b2)
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        self.p.addcdiv_(self.m, denom, value=-(self.lr / bc1))


def search_clip(seq, score_fn, lam1, lam2, N, mask_type='freeze', lr=0.2, init=None):
    """One clip's search, FindMasksComparison_I3D_smth.py:188-251.

    seq [1,C,T,H,W]; score_fn(x)->scalar tensor (differentiable).  No early exit
    (SURVEY.md F9).  Returns dict with trajectory [N,4] = (loss,l1,tv,score),
    final sigmoid mask, freeze_score (= last forward's score, :231,249),
    reverse_score (:234-235) and the init info."""
    if init is None:
        with torch.no_grad():
            tm, info = init_mask_central(seq, score_fn, 0.9, mask_type)
    else:
        tm, info = init.clone(), {}
    tm = tm.clone().requires_grad_()
    opt = Adam(tm.data, lr=lr)
    traj = []
    score = None
    for _ in range(N):
        if tm.grad is not None:
            tm.grad = None
        mc = torch.sigmoid(tm)                                 # :198
        l1 = lam1 * torch.sum(torch.abs(mc))                   # :199
        tv = lam2 * calc_tv_norm(mc, 3, 3)                     # :200
        score = score_fn(perturb_sequence(seq, mc, mask_type))  # :202-205
        loss = l1 + tv + score                                 # :207
        loss.backward()                                        # :213
        traj.append([loss.item(), l1.item(), tv.item(), score.item()])
        opt.step(tm.grad)                                      # :214
    final = torch.sigmoid(tm.detach())                         # :216
    with torch.no_grad():
        rev = float(score_fn(perturb_sequence(seq, final, 'reverse')))  # :234-235
    return dict(traj=torch.tensor(traj), mask=final, raw_mask=tm.detach().clone(),
                freeze_score=float(score.detach()) if score is not None else float('nan'),
                reverse_score=rev, init=info)


def frame_ranking(mask):
    """Integer frame-importance ranking: stable argsort of -mask."""
    return torch.argsort(-mask, stable=True)
