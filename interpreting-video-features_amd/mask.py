"""Drop-in for the reference's `mask` module (video_features_pytorch/mask.py) on
the MI355X: same function names, arguments and return conventions, every
computation a libivf_hip kernel (csrc/mask_ops.hip).  Tensors must live on the
GPU; there is no CPU path.

Both naming schemes found in the reference's drivers are accepted:
`perturbSequence`/`perbType=`, `calc_TVNorm`, `init_mask(thresh=, maskPertType=)`.
"""
import torch

import ivf_lib as L


class _FreezeFn(torch.autograd.Function):
    """mask.py:11-22 with its autograd (reverse scan) as explicit HIP kernels."""

    @staticmethod
    def forward(ctx, seq, mask):
        seq_c = L.f32c(seq)
        mask_c = L.f32c(mask)
        B, C, T, H, W = seq_c.shape
        per_clip = 1 if mask_c.dim() == 2 else 0
        out = torch.empty_like(seq_c)
        with torch.cuda.device(seq_c.device):
            L.check(L.lib().ivf_freeze_fwd(L.ptr(seq_c), L.ptr(mask_c), L.ptr(out), B, C, T, H * W, per_clip, 0,
                                           L.stream()))
        ctx.save_for_backward(seq_c, mask_c)
        ctx.per_clip = per_clip
        return out

    @staticmethod
    def backward(ctx, g):
        seq_c, mask_c = ctx.saved_tensors
        B, C, T, H, W = seq_c.shape
        g = L.f32c(g)
        ws = torch.empty(L.lib().ivf_freeze_bwd_workspace_bytes(B, T), dtype=torch.uint8, device=g.device)
        dm = torch.empty(B, T, device=g.device)
        dx = torch.empty_like(seq_c) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(g.device):
            L.check(L.lib().ivf_freeze_bwd(L.ptr(seq_c), L.ptr(mask_c), L.ptr(g), L.ptr(dm), L.ptr(dx), B, C, T,
                                           H * W, ctx.per_clip, 0, L.ptr(ws), L.stream()))
        if not ctx.per_clip:
            dm = dm.sum(dim=0)          # one mask shared by the batch (mask.py broadcasts it)
        return dx, dm


class _TVNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, p, q):
        m = L.f32c(mask).reshape(1, -1)
        T = m.shape[1]
        val = torch.empty(1, device=m.device)
        grad = torch.empty(1, T, device=m.device)
        with torch.cuda.device(m.device):
            L.check(L.lib().ivf_tv_norm(L.ptr(m), 1, T, float(p), float(q), L.ptr(val), L.ptr(grad), L.stream()))
        ctx.save_for_backward(grad)
        ctx.shape = mask.shape
        return val.reshape(())

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return (g * grad).reshape(ctx.shape), None, None


def _snap_in_place(mask):
    # mask.py:5-10 mutates the CALLER's tensor
    with torch.no_grad():
        mask.copy_((mask > 0.5).to(mask.dtype))


def find_submasks_from_mask(mask, thresh=0.1):
    """mask.py:60-85: list of index lists of consecutive frames with mask > thresh."""
    import ivf_search
    L.require_gpu(mask)
    return ivf_search.find_submasks_host(mask, thresh)


def _reverse(seq, mask, thresh=0.1):
    seq_c = L.f32c(seq)
    m = L.f32c(mask.detach()).reshape(-1)
    B, C, T, H, W = seq_c.shape
    run = torch.empty(T, dtype=torch.int32, device=seq_c.device)
    partner = torch.empty(T, dtype=torch.int32, device=seq_c.device)
    weight = torch.empty(T, device=seq_c.device)
    out = torch.empty_like(seq_c)
    with torch.cuda.device(seq_c.device):
        L.check(L.lib().ivf_submask_pairs(L.ptr(m), T, thresh, L.ptr(run), L.ptr(partner), L.ptr(weight), L.stream()))
        L.check(L.lib().ivf_reverse_fwd(L.ptr(seq_c), L.ptr(partner), L.ptr(weight), L.ptr(out), B, C, T, H * W, 0,
                                        L.stream()))
    return out


def perturb_sequence(seq, mask, perturbation_type='freeze', snap_values=False, perbType=None):
    """mask.py:4-57.  seq [B,C,T,H,W]; mask [T] (or [B,T] for per-clip masks, an
    extension).  'freeze' is differentiable w.r.t. mask and seq; 'reverse' is
    forward-only (the reference never back-propagates through it: smth:234-235)."""
    if perbType is not None:                      # legacy keyword, KTH:259,364
        perturbation_type = perbType
    L.require_gpu(seq, mask)
    if snap_values:
        _snap_in_place(mask)
    if perturbation_type == 'freeze':
        return _FreezeFn.apply(seq, mask)
    if perturbation_type == 'reverse':
        return _reverse(seq, mask, 0.1)
    # the reference falls through and hits an unbound local at its `return` (mask.py:57)
    raise UnboundLocalError("local variable 'perturbed_input' referenced before assignment")


def calc_tv_norm(mask, p=3, q=3):
    """mask.py:88-100, differentiable (NaN gradient when all entries are equal, as in
    the reference)."""
    L.require_gpu(mask)
    return _TVNormFn.apply(mask, p, q)


def init_mask(seq, model, batch_index, target, threshold=0.9, mode='central', mask_type='freeze',
              thresh=None, maskPertType=None):
    """mask.py:103-169.  `model` is any callable returning [B,K] scores (e.g.
    models.I3D_doubled.Model).  Returns the raw mask [T] (+-5 / +-2.5) with
    requires_grad set, on seq's device."""
    if thresh is not None:                        # legacy keywords, KTH:246-247
        threshold = thresh
    if maskPertType is not None:
        mask_type = maskPertType
    L.require_gpu(seq)
    dev = seq.device
    T = seq.shape[2]
    tgt = target[batch_index]
    with torch.no_grad():
        if mode == "central":
            fully_frozen = perturb_sequence(seq, torch.ones(T, device=dev), 'freeze')      # :123-126
            full = model(fully_frozen)[batch_index, tgt]                                   # :128
            orig = model(seq)[batch_index, tgt]                                            # :129
            new_mask = torch.ones(T, device=dev)
            for i in range(1, T // 2):                                                     # :134
                new_mask = torch.ones(T, device=dev)
                new_mask[:i] = 0
                new_mask[T - i:] = 0
                cen = model(perturb_sequence(seq, new_mask, perturbation_type=mask_type))[batch_index, tgt]
                ratio = (orig - cen) / (orig - full)                                       # :142
                if ratio < threshold:                                                      # :143
                    break
            m = torch.where(new_mask == 0, torch.tensor(-5.0, device=dev), torch.tensor(5.0, device=dev))
        elif mode == "random":
            m = (torch.rand(T, device=dev) > 0.7).float()                                  # :158
            m = (m - 0.5) * 5                                                              # :159-161
            if torch.abs(m.sum()) == 2.5 * T:                                              # :164-165
                m[8] += 0.1
        else:
            raise UnboundLocalError("local variable 'mask' referenced before assignment")
    m = m.contiguous()
    m.requires_grad_()
    print("initial mask is: ", m)                                                          # :168
    return m


# names used by FindMasksComparison_I3D_KTH.py:257,259,364
calc_TVNorm = calc_tv_norm
perturbSequence = perturb_sequence
