"""Host-side owner of one libivf_hip network plan (I3D or CLSTM_4).

Holds the two device arenas (weights, workspace) as torch tensors -- PyTorch is
the allocator, nothing more -- and exposes the plan's entry points with torch
tensors in/out.  Everything that computes is in csrc/*.hip.
"""
import ctypes
import os
from ctypes import byref, c_char, c_int, c_void_p

import numpy as np
import torch

import ivf_arch as arch
import ivf_lib as L


# arithmetic of the Unit3D convolutions unless a caller overrides it (see include/ivf_hip.h)
DEFAULT_MATH = os.environ.get("IVF_MATH", "bf16x6")   # fp32-class: 3-way bf16 split, 6 MFMA passes
# per-layer kernel autotuning when weights are first loaded (IVF_AUTOTUNE=0: built-in heuristic)
AUTOTUNE = os.environ.get("IVF_AUTOTUNE", "1") != "0"


def _mode_id(mode):
    """perturbation type -> the C-ABI's mode id; anything else fails as the reference's
    perturb_sequence does (mask.py:57 returns an unset local)."""
    if mode == "freeze":
        return 0
    if mode == "reverse":
        return 1
    raise UnboundLocalError("local variable 'perturbed_input' referenced before assignment")


def _arena(nbytes, device):
    # torch's caching allocator returns >=512-byte aligned blocks
    t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    assert t.data_ptr() % 256 == 0
    return t


class I3DEngine:
    """Plan + arenas for `models.I3D_doubled{,_kth}.Model` on `max_batch` clips of
    geometry [C,T,H,W] (reference Model.forward, I3D_doubled.py:351-380)."""

    def __init__(self, num_classes, clip_shape, max_batch=1, stride_mod_layers="", last_stride=1,
                 head_hw=(7, 7), head_time_base=2, softmax=True, device=None, math=None):
        L.require_gpu()
        self.device = torch.device(device if device is not None else "cuda")
        C, T, H, W = clip_shape
        sml = arch.parse_stride_mod(stride_mod_layers)
        cfg = L.I3DConfig()
        cfg.B, cfg.C, cfg.T, cfg.H, cfg.W = int(max_batch), C, T, H, W
        cfg.num_classes = int(num_classes)
        cfg.stem_stride_t = arch.temporal_stride('Conv3d_1a_7x7', sml, last_stride)
        cfg.pool4a_stride_t = arch.temporal_stride('MaxPool3d_4a_3x3', sml, last_stride)
        cfg.pool5a_stride_t = arch.temporal_stride('MaxPool3d_5a_2x2', sml, last_stride)
        cfg.head_kt = arch.head_time_kernel(sml, last_stride, head_time_base)
        cfg.head_kh, cfg.head_kw = head_hw
        cfg.softmax = 1 if softmax else 0
        self.math = math if math is not None else DEFAULT_MATH
        cfg.math = L.MATH_MODES[self.math]
        self.cfg = cfg
        self.clip_shape = (C, T, H, W)
        self.max_batch = int(max_batch)
        self.K = int(num_classes)
        self._h = c_void_p()
        L.check(L.lib().ivf_i3d_create(byref(cfg), byref(self._h)))
        with torch.cuda.device(self.device):
            self._weights = _arena(L.lib().ivf_i3d_weights_bytes(self._h), self.device)
            self._ws = _arena(L.lib().ivf_i3d_workspace_bytes(self._h), self.device)
        L.check(L.lib().ivf_i3d_bind(self._h, L.ptr(self._weights), L.ptr(self._ws)))
        self._tuned = False
        self.unit_names = []
        for i in range(L.lib().ivf_i3d_num_convs(self._h)):
            name = ctypes.create_string_buffer(64)
            L.check(L.lib().ivf_i3d_conv_info(self._h, i, name, None, None, None, None, None, None))
            self.unit_names.append(name.value.decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and L is not None and getattr(L, "_lib", None) is not None:
            L._lib.ivf_i3d_destroy(h)
            self._h = c_void_p()

    @property
    def workspace_bytes(self):
        return self._ws.numel()

    # -------------------------------------------------------------- weights
    def load_state_dict(self, sd, bn_eps=1e-3, autotune=None):
        """sd: reference key scheme, optional 'module.' prefix."""
        def get(key):
            for k in (key, "module." + key):
                if k in sd:
                    v = sd[k]
                    if isinstance(v, np.ndarray):
                        v = torch.from_numpy(v)
                    return L.f32c(v.detach().to(self.device))
            raise KeyError(f"state_dict is missing '{key}'")
        keep = []
        with torch.cuda.device(self.device):
            for i, name in enumerate(self.unit_names):
                w = get(f"{name}.conv3d.weight")
                if name == "logits":
                    bias = get("logits.conv3d.bias")
                    keep += [w, bias]
                    L.check(L.lib().ivf_i3d_load_conv(self._h, i, L.ptr(w), None, None, None, None,
                                                      L.ptr(bias), bn_eps, L.stream()))
                else:
                    g, b = get(f"{name}.bn.weight"), get(f"{name}.bn.bias")
                    m, v = get(f"{name}.bn.running_mean"), get(f"{name}.bn.running_var")
                    keep += [w, g, b, m, v]
                    L.check(L.lib().ivf_i3d_load_conv(self._h, i, L.ptr(w), L.ptr(g), L.ptr(b), L.ptr(m),
                                                      L.ptr(v), None, bn_eps, L.stream()))
            torch.cuda.current_stream().synchronize()   # sources may be freed after this
        if (AUTOTUNE if autotune is None else autotune) and not self._tuned:
            self.autotune()

    # -------------------------------------------------------------- kernel selection
    def autotune(self, reps=3, sample=None):
        """Time every kernel variant of every convolution at the plan's batch size and keep the
        fastest per layer and direction.  Results: see get_tuning / set_tuning.

        The candidates are timed on the plan's own activation / gradient buffers, so these are filled first by one
        forward + backward of `sample` (a [max_batch,C,T,H,W] clip batch; default: seeded noise in [-1, 1]): on an
        untouched (all-zero) workspace the matrix pipe toggles nothing, the chip holds a higher clock and the
        MFMA-dense tiles rank 16-19 % better than they run on data (MI355X_MICROARCH.md, DVFS give-back)."""
        with torch.cuda.device(self.device):
            if sample is None:
                g = torch.Generator(device="cpu").manual_seed(0)
                one = torch.rand((1,) + tuple(self.clip_shape), generator=g) * 2.0 - 1.0
                sample = one.to(self.device).expand(self.max_batch, *self.clip_shape).contiguous()
            probs = self.forward(sample)
            self.backward(sample.shape[0], target=self.argmax(probs), want_dx=False)
            del sample, probs
            L.check(L.lib().ivf_i3d_autotune(self._h, self.max_batch, int(reps), L.stream()))
            torch.cuda.current_stream().synchronize()
        self._tuned = True

    def set_overlap(self, on):
        """Run the HBM-bound branch of every Inception module on a side stream beside the 3x3x3 convs
        (ivf_i3d_set_overlap; off by default, bit-identical results)."""
        L.check(L.lib().ivf_i3d_set_overlap(self._h, int(bool(on))))

    def get_tuning(self):
        n = 2 * L.lib().ivf_i3d_num_conv_ops(self._h)
        arr = (c_int * n)()
        L.check(L.lib().ivf_i3d_get_tuning(self._h, arr))
        return list(arr)

    def set_tuning(self, variants):
        n = 2 * L.lib().ivf_i3d_num_conv_ops(self._h)
        if len(variants) != n:
            raise L.IvfError(f"tuning vector must have {n} entries")
        arr = (c_int * n)(*[int(v) for v in variants])
        L.check(L.lib().ivf_i3d_set_tuning(self._h, arr))
        self._tuned = True

    # -------------------------------------------------------------- helpers
    def _clip(self, x):
        L.require_gpu(x)
        x = L.f32c(x)
        if x.dim() != 5 or tuple(x.shape[1:]) != self.clip_shape:
            raise L.IvfError(f"clip batch must be [b,{','.join(map(str, self.clip_shape))}], got {tuple(x.shape)}")
        if x.shape[0] > self.max_batch:
            raise L.IvfError(f"batch {x.shape[0]} exceeds the plan's maximum {self.max_batch}")
        return x

    def _targets(self, target, b):
        t = torch.as_tensor(target, device=self.device).to(torch.int32).reshape(-1).contiguous()
        if t.numel() != b:
            raise L.IvfError(f"need {b} targets, got {t.numel()}")
        return t

    # -------------------------------------------------------------- entry points
    def forward(self, x, want_logits=False):
        x = self._clip(x)
        b = x.shape[0]
        probs = torch.empty(b, self.K, device=self.device)
        logits = torch.empty(b, self.K, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_i3d_forward(self._h, L.ptr(x), b, L.ptr(logits), L.ptr(probs), L.stream()))
        return (probs, logits) if want_logits else probs

    def backward(self, b, target=None, dout=None, want_dx=True):
        """Backward-data of the last forward.  Returns (score [b] or None, dx NCTHW or None)."""
        C, T, H, W = self.clip_shape
        tgt = self._targets(target, b) if target is not None else None
        dout = L.f32c(dout) if dout is not None else None
        score = torch.empty(b, device=self.device) if tgt is not None else None
        dx = torch.empty(b, C, T, H, W, device=self.device) if want_dx else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_i3d_backward(self._h, b, L.ptr(tgt), L.ptr(dout), L.ptr(score), L.ptr(dx),
                                             L.stream()))
        return score, dx

    def endpoint(self, name, b):
        """Activation of the last forward as an NCTHW torch tensor (copy)."""
        p = c_void_p()
        T, H, W, C, ld = c_int(), c_int(), c_int(), c_int(), c_int()
        L.check(L.lib().ivf_i3d_endpoint(self._h, name.encode(), byref(p), byref(T), byref(H), byref(W),
                                         byref(C), byref(ld)))
        off = p.value - self._ws.data_ptr()
        n = b * T.value * H.value * W.value * ld.value
        esz = L.lib().ivf_i3d_act_elem_bytes(self._h)           # 2: bf16 storage (math="bf16act")
        flat = self._ws[off:off + esz * n].view(torch.float32 if esz == 4 else torch.bfloat16).float()
        return flat.view(b, T.value, H.value, W.value, ld.value)[..., :C.value].permute(0, 4, 1, 2, 3).contiguous()

    def search(self, x, target, raw_mask, lam1, lam2, N, lr=0.2, betas=(0.9, 0.999), eps=1e-8,
               state=None, want_traj=True, mode="freeze"):
        """N iterations of the hot loop (smth:193-214) on b clips.  raw_mask [b,T] is
        updated in place; state = (exp_avg, exp_avg_sq, steps_done) continues a search;
        mode = the perturbation the loop optimises through (temporalMaskType, smth:121,202)."""
        x = self._clip(x)
        b = x.shape[0]
        T = self.clip_shape[1]
        tgt = self._targets(target, b)
        L.require_gpu(raw_mask)
        if raw_mask.dtype != torch.float32 or not raw_mask.is_contiguous() or tuple(raw_mask.shape) != (b, T):
            raise L.IvfError("raw_mask must be a contiguous float32 [b,T] tensor")
        if state is None:
            state = (torch.zeros_like(raw_mask), torch.zeros_like(raw_mask), 0)
        m, v, done = state
        traj = torch.empty(N, b, 4, device=self.device) if want_traj else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_i3d_search(self._h, L.ptr(x), b, L.ptr(tgt), L.ptr(raw_mask), L.ptr(m), L.ptr(v),
                                           lam1, lam2, lr, betas[0], betas[1], eps, int(N), done + 1,
                                           _mode_id(mode), L.ptr(traj), L.stream()))
        return traj, (m, v, done + int(N))

    def perturbed_forward(self, x, mask, mode="freeze"):
        x = self._clip(x)
        b = x.shape[0]
        mask = L.f32c(mask.to(self.device))
        if tuple(mask.shape) != (b, self.clip_shape[1]):
            raise L.IvfError("mask must be [b,T]")
        probs = torch.empty(b, self.K, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_i3d_perturbed_forward(self._h, L.ptr(x), b, L.ptr(mask),
                                                      _mode_id(mode), L.ptr(probs), L.stream()))
        return probs

    def argmax(self, probs):
        b = probs.shape[0]
        t = torch.empty(b, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_argmax(L.ptr(L.f32c(probs)), b, probs.shape[1], L.ptr(t), L.stream()))
        return t

    def gradcam(self, x, target=None, per_frame=True, out_hw=None, layer="Mixed_5c"):
        """GradCamVideo for b clips: (cam [b,T,H,W], probs [b,K]).  `layer`: the target endpoint
        (Conv3d_1a_7x7 ... Mixed_5c; the reference drivers use Mixed_5c, smth:258)."""
        x = self._clip(x)
        b = x.shape[0]
        C, T, H, W = self.clip_shape
        oh, ow = out_hw if out_hw is not None else (H, W)
        if target is None:
            target = self.argmax(self.forward(x))
        tgt = self._targets(target, b)
        p = c_void_p()
        Tf = c_int()
        if "." in layer or layer == "input":
            raise L.IvfError(f"'{layer}' is not an endpoint of the model")
        L.check(L.lib().ivf_i3d_endpoint(self._h, layer.encode(), byref(p), byref(Tf), None, None, None, None))
        frames = Tf.value * (T // Tf.value)
        cam = torch.empty(b, frames, oh, ow, device=self.device)
        probs = torch.empty(b, self.K, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_i3d_gradcam_layer(self._h, L.ptr(x), b, L.ptr(tgt), layer.encode(),
                                                  1 if per_frame else 0, oh, ow, L.ptr(cam), L.ptr(probs), L.stream()))
        return cam, probs


class CLSTMEngine:
    """Plan + arenas for `models.CLSTM_4.Model` (reference CLSTM_4.py:69-85 over
    convolution_lstm.py:96-132) on `max_batch` clips [C,T,H,W]."""

    def __init__(self, num_classes, clip_shape, max_batch=1, hidden=4, layers=2, kernel=5, stride=2,
                 softmax=False, batch_norm=True, out_step=None, out_steps=None, device=None):
        L.require_gpu()
        self.device = torch.device(device if device is not None else "cuda")
        C, T, H, W = clip_shape
        cfg = L.CLSTMConfig()
        cfg.B, cfg.C, cfg.T, cfg.H, cfg.W = int(max_batch), C, T, H, W
        cfg.hidden, cfg.layers, cfg.kernel, cfg.stride = int(hidden), int(layers), int(kernel), int(stride)
        cfg.num_classes = int(num_classes)
        cfg.softmax = 1 if softmax else 0
        cfg.batch_norm = 1 if batch_norm else 0
        cfg.out_step = T - 1 if out_step is None else int(out_step)
        if out_steps is not None:           # use_entire_seq: every effective step reached feeds endFC
            if not 1 <= len(out_steps) <= 16:
                raise L.IvfError("between 1 and 16 output steps")
            cfg.n_out_steps = len(out_steps)
            for i, sv in enumerate(out_steps):
                cfg.out_steps[i] = int(sv)
            cfg.out_step = int(out_steps[-1])
        self.cfg = cfg
        self.clip_shape = (C, T, H, W)
        self.max_batch = int(max_batch)
        self.K = int(num_classes)
        self.layers = int(layers)
        self._h = c_void_p()
        L.check(L.lib().ivf_clstm_create(byref(cfg), byref(self._h)))
        with torch.cuda.device(self.device):
            self._weights = _arena(L.lib().ivf_clstm_weights_bytes(self._h), self.device)
            self._ws = _arena(L.lib().ivf_clstm_workspace_bytes(self._h), self.device)
        L.check(L.lib().ivf_clstm_bind(self._h, L.ptr(self._weights), L.ptr(self._ws)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and L is not None and getattr(L, "_lib", None) is not None:
            L._lib.ivf_clstm_destroy(h)
            self._h = c_void_p()

    def load_state_dict(self, sd, bn_eps=1e-5):
        def get(key):
            for k in (key, "module." + key):
                if k in sd:
                    v = sd[k]
                    if isinstance(v, np.ndarray):
                        v = torch.from_numpy(v)
                    return L.f32c(v.detach().to(self.device))
            raise KeyError(f"state_dict is missing '{key}'")
        with torch.cuda.device(self.device):
            for i in range(self.layers):
                ts = ([get(f"clstm.cell{i}.Wx{g}.weight") for g in "ifco"]
                      + [get(f"clstm.cell{i}.Wx{g}.bias") for g in "ifco"]
                      + [get(f"clstm.cell{i}.Wh{g}.weight") for g in "ifco"])
                L.check(L.lib().ivf_clstm_load_cell(self._h, i, *[L.ptr(t) for t in ts], L.stream()))
                torch.cuda.current_stream().synchronize()
            bn = [get(f"clstm.bn.{k}") for k in ("weight", "bias", "running_mean", "running_var")] \
                if self.cfg.batch_norm else [None] * 4
            fw, fb = get("endFC.weight"), get("endFC.bias")
            L.check(L.lib().ivf_clstm_load_head(self._h, *[L.ptr(t) for t in bn], L.ptr(fw), L.ptr(fb), bn_eps,
                                                L.stream()))
            torch.cuda.current_stream().synchronize()

    _clip = I3DEngine._clip
    _targets = I3DEngine._targets
    argmax = I3DEngine.argmax

    def forward(self, x, want_logits=False):
        x = self._clip(x)
        b = x.shape[0]
        probs = torch.empty(b, self.K, device=self.device)
        logits = torch.empty(b, self.K, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_clstm_forward(self._h, L.ptr(x), b, L.ptr(logits), L.ptr(probs), L.stream()))
        return (probs, logits) if want_logits else probs

    def backward(self, b, target=None, dout=None, want_dx=True):
        C, T, H, W = self.clip_shape
        tgt = self._targets(target, b) if target is not None else None
        dout = L.f32c(dout) if dout is not None else None
        score = torch.empty(b, device=self.device) if tgt is not None else None
        dx = torch.empty(b, C, T, H, W, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_clstm_backward(self._h, b, L.ptr(tgt), L.ptr(dout), L.ptr(score), L.ptr(dx),
                                               L.stream()))
        return score, dx

    def search(self, x, target, raw_mask, lam1, lam2, N, lr=0.2, betas=(0.9, 0.999), eps=1e-8,
               state=None, want_traj=True, mode="freeze"):
        x = self._clip(x)
        b = x.shape[0]
        T = self.clip_shape[1]
        tgt = self._targets(target, b)
        L.require_gpu(raw_mask)
        if raw_mask.dtype != torch.float32 or not raw_mask.is_contiguous() or tuple(raw_mask.shape) != (b, T):
            raise L.IvfError("raw_mask must be a contiguous float32 [b,T] tensor")
        if state is None:
            state = (torch.zeros_like(raw_mask), torch.zeros_like(raw_mask), 0)
        m, v, done = state
        traj = torch.empty(N, b, 4, device=self.device) if want_traj else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_clstm_search(self._h, L.ptr(x), b, L.ptr(tgt), L.ptr(raw_mask), L.ptr(m), L.ptr(v),
                                             lam1, lam2, lr, betas[0], betas[1], eps, int(N), done + 1,
                                             _mode_id(mode), L.ptr(traj), L.stream()))
        return traj, (m, v, done + int(N))

    def perturbed_forward(self, x, mask, mode="freeze"):
        x = self._clip(x)
        b = x.shape[0]
        mask = L.f32c(mask.to(self.device))
        if tuple(mask.shape) != (b, self.clip_shape[1]):
            raise L.IvfError("mask must be [b,T]")
        probs = torch.empty(b, self.K, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_clstm_perturbed_forward(self._h, L.ptr(x), b, L.ptr(mask),
                                                        _mode_id(mode), L.ptr(probs), L.stream()))
        return probs

    def gradcam(self, *a, **kw):
        raise L.IvfError("Grad-CAM for the PyTorch ConvLSTM cannot run in the reference either "
                         "(grad-cam.py:33-49 refers to attributes CLSTM_4.Model lacks, SURVEY.md F5); "
                         "not built")


class TFCLSTMEngine:
    """SURVEY 8f N4, a DOCUMENTED EXTENSION (parity unpinned): the TF half's Keras ConvLSTM2D classifier
    (video_features_tf/models/clstm.py:87-126) and its temporal-mask search / per-frame Grad-CAM
    (mask/find_mask_kth.py:300-452, mask/gradcam.py:28-111) on libivf_hip's generic direct-convolution kernels
    (csrc/tf_clstm.hip).  Clips are NCTHW like everywhere else; `from_tf_layout` converts [B,T,H,W,C]."""

    def __init__(self, num_classes, clip_shape, units=(32, 32), kernel=(3, 5), stride=2, padding="valid",
                 recurrent_activation="hard_sigmoid", only_last_element_for_fc=True, max_batch=1, device=None):
        L.require_gpu()
        self.device = torch.device(device if device is not None else "cuda")
        C, T, H, W = clip_shape
        cfg = L.TFCLSTMConfig()
        cfg.B, cfg.C, cfg.T, cfg.H, cfg.W = int(max_batch), C, T, H, W
        cfg.layers = len(units)
        for i, u in enumerate(units):
            cfg.units[i] = int(u)
        cfg.kh, cfg.kw = int(kernel[0]), int(kernel[1])
        cfg.stride = int(stride)
        if padding not in ("valid", "same"):
            raise L.IvfError("padding must be 'valid' or 'same'")
        cfg.padding = 1 if padding == "same" else 0
        if recurrent_activation not in ("hard_sigmoid", "sigmoid"):
            raise L.IvfError("recurrent_activation must be 'hard_sigmoid' or 'sigmoid'")
        cfg.recurrent_hard_sigmoid = 1 if recurrent_activation == "hard_sigmoid" else 0
        cfg.only_last = 1 if only_last_element_for_fc else 0
        cfg.num_classes = int(num_classes)
        self.cfg, self.clip_shape, self.max_batch, self.K = cfg, (C, T, H, W), int(max_batch), int(num_classes)
        self.units = tuple(int(u) for u in units)
        self._h = c_void_p()
        L.check(L.lib().ivf_tfclstm_create(byref(cfg), byref(self._h)))
        with torch.cuda.device(self.device):
            self._weights = _arena(L.lib().ivf_tfclstm_weights_bytes(self._h), self.device)
            self._ws = _arena(L.lib().ivf_tfclstm_workspace_bytes(self._h), self.device)
        L.check(L.lib().ivf_tfclstm_bind(self._h, L.ptr(self._weights), L.ptr(self._ws)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and L is not None and getattr(L, "_lib", None) is not None:
            L._lib.ivf_tfclstm_destroy(h)
            self._h = c_void_p()

    @staticmethod
    def from_tf_layout(x_bthwc):
        """[B,T,H,W,C] (the TF graph's sequence layout) -> NCTHW."""
        return x_bthwc.permute(0, 4, 1, 2, 3).contiguous()

    @property
    def fc_inputs(self):
        return L.lib().ivf_tfclstm_fc_inputs(self._h)

    def layer_dims(self, i):
        v = [c_int() for _ in range(5)]
        L.check(L.lib().ivf_tfclstm_layer_dims(self._h, i, *[byref(a) for a in v]))
        return tuple(a.value for a in v)

    def load_weights(self, layers, dense_w, dense_b):
        """layers: [(kernel [kh,kw,Cin,4F], recurrent_kernel [kh,kw,F,4F], bias [4F])] in Keras layouts;
        dense_w [inputs, classes], dense_b [classes]."""
        dev = self.device
        with torch.cuda.device(dev):
            for i, (k, rk, b) in enumerate(layers):
                k, rk, b = (L.f32c(torch.as_tensor(t).to(dev)) for t in (k, rk, b))
                L.check(L.lib().ivf_tfclstm_load_layer(self._h, i, L.ptr(k), L.ptr(rk), L.ptr(b), L.stream()))
                torch.cuda.current_stream().synchronize()
            dw, db = L.f32c(torch.as_tensor(dense_w).to(dev)), L.f32c(torch.as_tensor(dense_b).to(dev))
            if tuple(dw.shape) != (self.fc_inputs, self.K):
                raise L.IvfError(f"dense kernel must be [{self.fc_inputs},{self.K}], got {tuple(dw.shape)}")
            L.check(L.lib().ivf_tfclstm_load_head(self._h, L.ptr(dw), L.ptr(db), L.stream()))
            torch.cuda.current_stream().synchronize()

    _clip = I3DEngine._clip
    _targets = I3DEngine._targets

    def forward(self, x, want_logits=False):
        x = self._clip(x)
        b = x.shape[0]
        probs = torch.empty(b, self.K, device=self.device)
        logits = torch.empty(b, self.K, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_tfclstm_forward(self._h, L.ptr(x), b, L.ptr(logits), L.ptr(probs), L.stream()))
        return (probs, logits) if want_logits else probs

    def backward(self, b, target):
        C, T, H, W = self.clip_shape
        tgt = self._targets(target, b)
        score = torch.empty(b, device=self.device)
        dx = torch.empty(b, C, T, H, W, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_tfclstm_backward(self._h, b, L.ptr(tgt), L.ptr(score), L.ptr(dx), L.stream()))
        return score, dx

    def perturbed_forward(self, x, mask):
        x = self._clip(x)
        b = x.shape[0]
        mask = L.f32c(mask.to(self.device))
        probs = torch.empty(b, self.K, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_tfclstm_perturbed_forward(self._h, L.ptr(x), b, L.ptr(mask), L.ptr(probs), L.stream()))
        return probs

    def search(self, x, target, raw_mask, lam1, lam2, N, lr=0.2, betas=(0.9, 0.999), eps=1e-8, state=None):
        x = self._clip(x)
        b, T = x.shape[0], self.clip_shape[1]
        tgt = self._targets(target, b)
        if raw_mask.dtype != torch.float32 or not raw_mask.is_contiguous() or tuple(raw_mask.shape) != (b, T):
            raise L.IvfError("raw_mask must be a contiguous float32 [b,T] tensor")
        if state is None:
            state = (torch.zeros_like(raw_mask), torch.zeros_like(raw_mask), 0)
        m, v, done = state
        traj = torch.empty(N, b, 4, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_tfclstm_search(self._h, L.ptr(x), b, L.ptr(tgt), L.ptr(raw_mask), L.ptr(m), L.ptr(v), lam1, lam2,
                                               lr, betas[0], betas[1], eps, int(N), done + 1, L.ptr(traj), L.stream()))
        return traj, (m, v, done + int(N))

    def gradcam(self, x, target, mask=None, normalization_mode="frame", out_hw=None):
        """mask/gradcam.py: per-frame maps [b,T,H,W]; normalization_mode 'frame' | 'sequence' (FLAGS.normalization_mode)."""
        x = self._clip(x)
        b = x.shape[0]
        C, T, H, W = self.clip_shape
        oh, ow = out_hw if out_hw is not None else (H, W)
        if normalization_mode not in ("frame", "sequence"):
            raise L.IvfError("Error. Need to provide normalization mode.")           # gradcam.py:97
        tgt = self._targets(target, b)
        m = L.f32c(mask.to(self.device)) if mask is not None else None
        cam = torch.empty(b, T, oh, ow, device=self.device)
        probs = torch.empty(b, self.K, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().ivf_tfclstm_gradcam(self._h, L.ptr(x), b, L.ptr(m), L.ptr(tgt), 1 if normalization_mode == "frame" else 0,
                                                oh, ow, L.ptr(cam), L.ptr(probs), L.stream()))
        return cam, probs
