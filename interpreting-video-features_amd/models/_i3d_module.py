"""Shared implementation of the drop-in I3D `Model` classes.

The torch.nn modules below are PARAMETER CONTAINERS only: they give the model the
reference's state_dict key scheme, default initialisation order and .to()/.cuda()
behaviour.  They are never called -- forward/backward run the HIP plan
(csrc/i3d_net.hip) through ivf_engine.I3DEngine.
"""
import torch
import torch.nn as nn

import ivf_arch as arch
import ivf_lib as L


class Unit3D(nn.Module):
    """Parameter holder with the reference's attribute names (I3D_doubled.py:43-75)."""

    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1), padding=0,
                 activation_fn=None, use_batch_norm=True, use_bias=False, name='unit_3d'):
        super().__init__()
        self._output_channels = output_channels
        self._kernel_shape = kernel_shape
        self._stride = stride
        self._use_batch_norm = use_batch_norm
        self._activation_fn = activation_fn
        self._use_bias = use_bias
        self.name = name
        self.padding = padding
        self.conv3d = nn.Conv3d(in_channels, output_channels, kernel_size=kernel_shape, stride=stride,
                                padding=0, bias=use_bias)
        if use_batch_norm:
            self.bn = nn.BatchNorm3d(output_channels, eps=0.001, momentum=0.01)

    def forward(self, x):
        raise L.IvfError("Unit3D is executed inside the whole-network HIP plan; call the Model")


class MaxPool3dSamePadding(nn.Module):
    def __init__(self, kernel_size, stride, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        raise L.IvfError("MaxPool3dSamePadding is executed inside the whole-network HIP plan; call the Model")


class InceptionModule(nn.Module):
    def __init__(self, in_channels, out_channels, name):
        super().__init__()
        self.b0 = Unit3D(in_channels, out_channels[0], [1, 1, 1], name=name + '/Branch_0/Conv3d_0a_1x1')
        self.b1a = Unit3D(in_channels, out_channels[1], [1, 1, 1], name=name + '/Branch_1/Conv3d_0a_1x1')
        self.b1b = Unit3D(out_channels[1], out_channels[2], [3, 3, 3], name=name + '/Branch_1/Conv3d_0b_3x3')
        self.b2a = Unit3D(in_channels, out_channels[3], [1, 1, 1], name=name + '/Branch_2/Conv3d_0a_1x1')
        self.b2b = Unit3D(out_channels[3], out_channels[4], [3, 3, 3], name=name + '/Branch_2/Conv3d_0b_3x3')
        self.b3a = MaxPool3dSamePadding([3, 3, 3], (1, 1, 1), 0)
        self.b3b = Unit3D(in_channels, out_channels[5], [1, 1, 1], name=name + '/Branch_3/Conv3d_0b_1x1')
        self.name = name

    def forward(self, x):
        raise L.IvfError("InceptionModule is executed inside the whole-network HIP plan; call the Model")


class _NetFn(torch.autograd.Function):
    """model(x) with autograd: backward = the plan's backward-data (dL/dx only)."""

    @staticmethod
    def forward(ctx, x, model):
        eng = model._engine_for(x)
        out = eng.forward(x)
        ctx.model, ctx.shape, ctx.b = model, tuple(x.shape), x.shape[0]
        model._last_forward_token = token = object()
        ctx.token = token
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        if model._last_forward_token is not ctx.token:
            raise L.IvfError("backward() must follow the forward it belongs to: the HIP plan keeps the "
                             "activations of the LAST forward only")
        eng = model._engine_cache[ctx.shape[1:]][1]
        _, dx = eng.backward(ctx.b, dout=dout.contiguous(), want_dx=True)
        return dx, None


class I3DBase(nn.Module):
    VALID_ENDPOINTS = arch.ENDPOINTS + ('Logits', 'Predictions')
    _HEAD_HW = (7, 7)

    def _construct(self, num_classes, spatial_squeeze, final_endpoint, name, in_channels, dropout_keep_prob,
                   last_stride, stride_mod_layers, softMax, lastRelu, head_time_base):
        if final_endpoint not in self.VALID_ENDPOINTS:
            raise ValueError('Unknown final endpoint %s' % final_endpoint)
        if final_endpoint != 'Logits':
            raise L.IvfError("only final_endpoint='Logits' is built on the HIP path")
        self._num_classes = num_classes
        self._spatial_squeeze = spatial_squeeze
        self._final_endpoint = final_endpoint
        self.softMax = softMax
        self.lastRelu = lastRelu
        self._in_channels = in_channels
        self._last_stride = last_stride
        self._stride_mod_layers = arch.parse_stride_mod(stride_mod_layers)   # TypeError on None, as the reference
        self._head_time_base = head_time_base
        # registration order of the reference: sm, avg_pool, dropout, logits, then the
        # endpoints (I3D_doubled.py:216,313-334,346-349) -- keeps default init identical
        self.sm = nn.Softmax(dim=1)
        eps = {}
        st = arch.temporal_stride('Conv3d_1a_7x7', self._stride_mod_layers, last_stride)
        eps['Conv3d_1a_7x7'] = Unit3D(in_channels, 64, [7, 7, 7], (st, 2, 2), (3, 3, 3), name=name + 'Conv3d_1a_7x7')
        eps['MaxPool3d_2a_3x3'] = MaxPool3dSamePadding([1, 3, 3], (1, 2, 2))
        eps['Conv3d_2b_1x1'] = Unit3D(64, 64, [1, 1, 1], name=name + 'Conv3d_2b_1x1')
        eps['Conv3d_2c_3x3'] = Unit3D(64, 192, [3, 3, 3], padding=1, name=name + 'Conv3d_2c_3x3')
        eps['MaxPool3d_3a_3x3'] = MaxPool3dSamePadding([1, 3, 3], (1, 2, 2))
        for ep in arch.ENDPOINTS[5:]:
            if ep in arch.INCEPTION:
                cin, oc = arch.INCEPTION[ep]
                eps[ep] = InceptionModule(cin, oc, name + ep)
            else:
                k, s = arch.POOLS[ep]
                eps[ep] = MaxPool3dSamePadding(list(k), (arch.temporal_stride(ep, self._stride_mod_layers, last_stride),
                                                          s[1], s[2]))
        self.end_points = eps
        kt = arch.head_time_kernel(self._stride_mod_layers, last_stride, head_time_base)
        self.avg_pool = nn.AvgPool3d(kernel_size=[kt, self._HEAD_HW[0], self._HEAD_HW[1]], stride=(1, 1, 1))
        self.dropout = nn.Dropout(dropout_keep_prob)
        self.logits = Unit3D(arch.FEATURE_CHANNELS, num_classes, [1, 1, 1], use_batch_norm=False, use_bias=True,
                             name='logits')
        self.build()
        self._engine_cache = {}
        self._last_forward_token = None
        self._weights_version = 0

    def build(self):
        for k in self.end_points.keys():
            self.add_module(k, self.end_points[k])

    def replace_logits(self, num_classes):
        self._num_classes = num_classes
        self.logits = Unit3D(arch.FEATURE_CHANNELS, num_classes, [1, 1, 1], use_batch_norm=False, use_bias=True,
                             name='logits')
        self.refresh()

    # ---------------------------------------------------------------- engine plumbing
    def refresh(self):
        """Call after editing parameters in place: re-packs them into the HIP plan."""
        self._weights_version += 1

    def load_state_dict(self, state_dict, strict=True):
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        r = super().load_state_dict(sd, strict=strict)
        self.refresh()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        if hasattr(self, '_engine_cache'):
            self._engine_cache = {}
        return r

    def _engine_for(self, x, min_batch=1):
        import ivf_engine
        L.require_gpu(x)
        if self.training:
            raise L.IvfError("the HIP path implements eval-mode semantics (BatchNorm running statistics, "
                             "dropout off); call model.eval() first, as find_masks does (smth:145)")
        key = tuple(x.shape[1:])
        need = max(int(x.shape[0]), min_batch)
        ent = self._engine_cache.get(key)
        if ent is None or ent[1].max_batch < need or ent[1].device != x.device:
            eng = ivf_engine.I3DEngine(self._num_classes, key, max_batch=need,
                                       stride_mod_layers=self._stride_mod_layers, last_stride=self._last_stride,
                                       head_hw=self._HEAD_HW, head_time_base=self._head_time_base,
                                       softmax=bool(self.softMax), device=x.device)
            ent = [-1, eng]
            self._engine_cache[key] = ent
        if ent[0] != self._weights_version:
            ent[1].load_state_dict(self.state_dict())
            ent[0] = self._weights_version
        return ent[1]

    def forward(self, x):
        """I3D_doubled.py:351-380: [B,C,T,H,W] -> [B,num_classes] (softmax if softMax)."""
        if not self._spatial_squeeze:
            raise L.IvfError("spatial_squeeze=False is not built on the HIP path")
        if self.lastRelu == "relu":
            raise L.IvfError("lastRelu='relu' is not built on the HIP path (the reference drivers use None)")
        return _NetFn.apply(x, self)

    def extract_features(self, x):
        """I3D_doubled.py:382-388: the average-pooled Mixed_5c features [B,1024,1,1,1]."""
        eng = self._engine_for(x)
        eng.forward(x)
        return eng.endpoint('Mixed_5c', x.shape[0]).mean(dim=(2, 3, 4), keepdim=True)
