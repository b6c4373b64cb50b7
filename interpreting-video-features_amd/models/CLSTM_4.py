"""Drop-in for video_features_pytorch/models/CLSTM_4.py: same constructor and
state_dict keys (`clstm.bn.*`, `clstm.cell{i}.W{x,h}{i,f,c,o}.*`, `endFC.*`);
forward/backward run the ConvLSTM HIP plan (csrc/convlstm.hip)."""
import torch

import ivf_lib as L
from models.convolution_lstm import ConvLSTM


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model):
        eng = model._engine_for(x)
        out = eng.forward(x)
        ctx.model, ctx.b, ctx.key = model, x.shape[0], tuple(x.shape[1:])
        model._last_forward_token = ctx.token = object()
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        if model._last_forward_token is not ctx.token:
            raise L.IvfError("backward() must follow the forward it belongs to: the HIP plan keeps the "
                             "activations of the LAST forward only")
        _, dx = model._engine_cache[ctx.key][1].backward(ctx.b, dout=dout.contiguous())
        return dx, None


class Model(torch.nn.Module):
    def __init__(self, num_classes=174, nb_lstm_units=32, channels=3, conv_kernel_size=(5, 5), pool_kernel_size=(2, 2),
                 top_layer=True, avg_pool=False, batch_normalization=True, lstm_layers=4, step=16,
                 image_size=(224, 224), dropout=0, conv_stride=(1, 1), effective_step=[4, 8, 12, 15],
                 use_entire_seq=False, add_softmax=False):
        super().__init__()
        self.num_classes, self.nb_lstm_units, self.channels = num_classes, nb_lstm_units, channels
        self.top_layer, self.avg_pool = top_layer, avg_pool
        self.c_kernel_size = conv_kernel_size
        self.lstm_layers, self.step, self.im_size = lstm_layers, step, image_size
        self.pool_kernel_size = pool_kernel_size
        self.batch_normalization, self.dropout = batch_normalization, dropout
        self.conv_stride = conv_stride
        self.effective_step = effective_step
        self.add_softmax, self.use_entire_seq = add_softmax, use_entire_seq
        self.clstm = self.endFC = self.sm = None
        self.build()
        self._engine_cache = {}
        self._last_forward_token = None
        self._weights_version = 0

    def _feat(self):
        red = (self.conv_stride * self.pool_kernel_size[0]) ** self.lstm_layers     # CLSTM_4.py:60-62
        return self.nb_lstm_units * int(self.im_size[0] / red) * int(self.im_size[1] / red)

    def build(self):
        """CLSTM_4.py:38-67."""
        if isinstance(self.conv_stride, (tuple, list)):
            raise TypeError("conv_stride must be an int: the reference multiplies it by pool_kernel_size[0] "
                            "(CLSTM_4.py:60); its tuple default cannot run")
        self.clstm = ConvLSTM(input_channels=self.channels, hidden_channels=[self.nb_lstm_units] * self.lstm_layers,
                              kernel_size=self.c_kernel_size[0], conv_stride=self.conv_stride,
                              pool_kernel_size=self.pool_kernel_size, step=self.step,
                              effective_step=self.effective_step, batch_normalization=self.batch_normalization,
                              dropout=self.dropout)
        mult = len(self.effective_step) if self.use_entire_seq else 1
        self.endFC = torch.nn.Linear(in_features=mult * self._feat(), out_features=self.num_classes)
        print("use entire sequence is: ", self.use_entire_seq)
        print("shape of FC is: ", self.endFC)
        self.sm = torch.nn.Softmax(dim=1)

    def _out_step(self):
        """The reference classifies `output[-1]`: the LAST effective step the unrolled loop actually
        reached, i.e. the largest entry below `step` (convolution_lstm.py:129-130); with none the
        list is empty and `output[-1]` raises IndexError."""
        reached = [s for s in self.effective_step if 0 <= s < self.step]
        if not reached:
            raise IndexError("list index out of range")
        return max(reached)

    def _out_steps(self):
        """use_entire_seq (CLSTM_4.py:73-76): the outputs of every effective step reached, in step order.
        The reference's `torch.stack(output).view(-1, E*feat)` equals this per-clip concatenation for one
        clip; for B > 1 its view mixes the clips of a batch (rows are cut from the [E,B,feat] stack), here
        every row stays the clip's own concatenation, which is what one-clip calls of the reference give."""
        reached = sorted({s for s in self.effective_step if 0 <= s < self.step})
        if len(reached) != len(self.effective_step):
            raise RuntimeError("shape '[-1, %d]' is invalid: %d of the %d effective steps lie inside the %d-step clip"
                               % (len(self.effective_step) * self._feat(), len(reached), len(self.effective_step),
                                  self.step))
        return reached

    def refresh(self):
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
            raise L.IvfError("the HIP path implements eval-mode semantics; call model.eval() first")
        if self.nb_lstm_units > 4 and (self.nb_lstm_units > 32 or self.nb_lstm_units % 4):
            raise L.IvfError("the HIP ConvLSTM plan takes nb_lstm_units 1..4 or a multiple of 4 up to 32")
        key = tuple(x.shape[1:])
        if key[1] != self.step:
            raise L.IvfError(f"clip has {key[1]} frames, model was built with step={self.step}")
        need = max(int(x.shape[0]), min_batch)
        ent = self._engine_cache.get(key)
        if ent is None or ent[1].max_batch < need or ent[1].device != x.device:
            eng = ivf_engine.CLSTMEngine(self.num_classes, key, max_batch=need, hidden=self.nb_lstm_units,
                                         layers=self.lstm_layers, kernel=self.c_kernel_size[0],
                                         stride=self.conv_stride, softmax=bool(self.add_softmax),
                                         batch_norm=bool(self.batch_normalization),
                                         out_step=self._out_step(),
                                         out_steps=self._out_steps() if self.use_entire_seq else None,
                                         device=x.device)
            ent = [-1, eng]
            self._engine_cache[key] = ent
        if ent[0] != self._weights_version:
            ent[1].load_state_dict(self.state_dict())
            ent[0] = self._weights_version
        return ent[1]

    def forward(self, x):
        """CLSTM_4.py:69-85: [B,C,T,H,W] -> [B,num_classes]."""
        return _NetFn.apply(x, self)
