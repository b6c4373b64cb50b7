"""Oracle (test infrastructure): CLSTM_4 classifier forward, functional torch CPU.

Follows video_features_pytorch/models/convolution_lstm.py (cell :38-48,
init_hidden :50-60, unrolled stack :96-132) and models/CLSTM_4.py:69-85.
"""
import torch
import torch.nn.functional as F


def cell(x, h, c, sd, prefix, stride, pad):
    """convolution_lstm.py:38-48 with the peephole terms dropped: Wci/Wcf/Wco are
    zero non-parameters (:52-54) so `c * W` contributes exactly 0."""
    def gate(g):
        return (F.conv2d(x, sd[f'{prefix}.Wx{g}.weight'], sd[f'{prefix}.Wx{g}.bias'], stride, pad)
                + F.conv2d(h, sd[f'{prefix}.Wh{g}.weight'], None, 1, pad))
    ci = torch.sigmoid(gate('i'))
    cf = torch.sigmoid(gate('f'))
    cc = cf * c + ci * torch.tanh(gate('c'))
    co = torch.sigmoid(gate('o'))
    return co * torch.tanh(cc), cc


def convlstm(x, sd, layers, hidden, kernel, stride, steps, effective_step, batch_norm=True, pool=2):
    """convolution_lstm.py:96-132.  x [B,C,T,H,W] -> list of pooled outputs at the
    effective steps.  ONE BatchNorm2d is shared by all layers (:85,123)."""
    pad = (kernel - 1) // 2
    state, outs = [], []
    for t in range(steps):
        v = x[:, :, t]
        for i in range(layers):
            if t == 0:
                b, _, hh, ww = v.shape
                z = v.new_zeros(b, hidden, hh // stride, ww // stride)
                state.append((z, z.clone()))
            h, c = state[i]
            v, c2 = cell(v, h, c, sd, f'clstm.cell{i}', stride, pad)
            state[i] = (v, c2)
            if batch_norm:
                v = F.batch_norm(v, sd['clstm.bn.running_mean'], sd['clstm.bn.running_var'],
                                 sd['clstm.bn.weight'], sd['clstm.bn.bias'], training=False, eps=1e-5)
            v = F.max_pool2d(v, pool)
        if t in effective_step:
            outs.append(v)
    return outs


def forward(x, sd, layers=2, hidden=4, kernel=5, stride=2, steps=32,
            effective_step=(7, 15, 23, 31), add_softmax=False, batch_norm=True, use_entire_seq=False):
    """CLSTM_4.py:69-85: endFC on the flattened LAST effective-step output, or (use_entire_seq,
    :73-76) on `torch.stack(output).view(-1, E*feat)` -- restated literally, so for B > 1 it mixes the
    clips of a batch exactly as the reference does; call it with one clip for per-clip semantics."""
    outs = convlstm(x, sd, layers, hidden, kernel, stride, steps, effective_step, batch_norm)
    if use_entire_seq:
        flat = torch.stack(outs).reshape(-1, len(outs) * outs[0][0].numel())
    else:
        flat = outs[-1].reshape(x.shape[0], -1)
    y = F.linear(flat, sd['endFC.weight'], sd['endFC.bias'])
    return torch.softmax(y, dim=1) if add_softmax else y
