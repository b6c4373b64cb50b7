"""Oracle (test infrastructure): Inception-v1 I3D forward, functional torch CPU
fp32 over a reference-layout state_dict.

Follows video_features_pytorch/models/I3D_doubled.py (Unit3D :83-118,
MaxPool3dSamePadding :8-40, InceptionModule :141-146, Model.forward :351-380)
and the KTH head of I3D_doubled_kth.py:302-308.
"""
import torch
import torch.nn.functional as F

_INCEPTION = ('Mixed_3b', 'Mixed_3c', 'Mixed_4b', 'Mixed_4c', 'Mixed_4d',
              'Mixed_4e', 'Mixed_4f', 'Mixed_5b', 'Mixed_5c')


def _same_pad(n, k, s):
    # I3D_doubled.py:9-13 / :77-81
    p = max(k - s, 0) if n % s == 0 else max(k - (n % s), 0)
    return p // 2, p - p // 2


def _pad5(x, kernel, stride):
    t, h, w = x.shape[2:]
    pt, ph, pw = (_same_pad(n, k, s) for n, k, s in zip((t, h, w), kernel, stride))
    return F.pad(x, (pw[0], pw[1], ph[0], ph[1], pt[0], pt[1]))  # :36-39 / :101-106


def unit3d(x, sd, prefix, stride=(1, 1, 1), bn=True, relu=True):
    """I3D_doubled.py:83-118: zero 'same' pad -> conv3d(no pad) -> BN(eps 1e-3,
    eval) -> ReLU.  The logits unit has bias, no BN, no activation (:327-334)."""
    w = sd[prefix + '.conv3d.weight']
    x = _pad5(x, w.shape[2:], stride)
    x = F.conv3d(x, w, sd.get(prefix + '.conv3d.bias'), stride=stride)
    if bn:
        x = F.batch_norm(x, sd[prefix + '.bn.running_mean'], sd[prefix + '.bn.running_var'],
                         sd[prefix + '.bn.weight'], sd[prefix + '.bn.bias'],
                         training=False, eps=1e-3)
    return F.relu(x) if relu else x


def maxpool_same(x, kernel, stride):
    """I3D_doubled.py:15-40: ZERO fill (not -inf), then MaxPool3d(ceil_mode=False)."""
    return F.max_pool3d(_pad5(x, kernel, stride), kernel, stride)


def inception(x, sd, name):
    """I3D_doubled.py:141-146."""
    b0 = unit3d(x, sd, name + '.b0')
    b1 = unit3d(unit3d(x, sd, name + '.b1a'), sd, name + '.b1b')
    b2 = unit3d(unit3d(x, sd, name + '.b2a'), sd, name + '.b2b')
    b3 = unit3d(maxpool_same(x, (3, 3, 3), (1, 1, 1)), sd, name + '.b3b')
    return torch.cat([b0, b1, b2, b3], dim=1)


def features(x, sd, stride_mod_layers="", last_stride=1, endpoints=None):
    """Everything up to and including Mixed_5c (I3D_doubled.py:353-357)."""
    def ts(ep):
        return last_stride if ep in stride_mod_layers else 2

    def rec(name, v):
        if endpoints is not None:
            endpoints[name] = v
        return v

    x = rec('Conv3d_1a_7x7', unit3d(x, sd, 'Conv3d_1a_7x7', (ts('Conv3d_1a_7x7'), 2, 2)))
    x = rec('MaxPool3d_2a_3x3', maxpool_same(x, (1, 3, 3), (1, 2, 2)))
    x = rec('Conv3d_2b_1x1', unit3d(x, sd, 'Conv3d_2b_1x1'))
    x = rec('Conv3d_2c_3x3', unit3d(x, sd, 'Conv3d_2c_3x3'))
    x = rec('MaxPool3d_3a_3x3', maxpool_same(x, (1, 3, 3), (1, 2, 2)))
    x = rec('Mixed_3b', inception(x, sd, 'Mixed_3b'))
    x = rec('Mixed_3c', inception(x, sd, 'Mixed_3c'))
    x = rec('MaxPool3d_4a_3x3', maxpool_same(x, (3, 3, 3), (ts('MaxPool3d_4a_3x3'), 2, 2)))
    for n in ('Mixed_4b', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e', 'Mixed_4f'):
        x = rec(n, inception(x, sd, n))
    x = rec('MaxPool3d_5a_2x2', maxpool_same(x, (2, 2, 2), (ts('MaxPool3d_5a_2x2'), 2, 2)))
    x = rec('Mixed_5b', inception(x, sd, 'Mixed_5b'))
    x = rec('Mixed_5c', inception(x, sd, 'Mixed_5c'))
    return x


def head(feat, sd, pool_kernel=(2, 7, 7), softmax=True):
    """I3D_doubled.py:360-380: AvgPool3d(stride 1) -> dropout(eval: identity) ->
    1x1x1 conv with bias -> squeeze(3).squeeze(3).squeeze() -> [None,:] if 1-D ->
    Softmax(dim=1).  Returns (pre-softmax logits, output)."""
    x = F.avg_pool3d(feat, pool_kernel, stride=(1, 1, 1))
    x = unit3d(x, sd, 'logits', bn=False, relu=False)
    logits = x.squeeze(3).squeeze(3).squeeze()
    if logits.dim() < 2:
        logits = logits[None, :]
    out = torch.softmax(logits, dim=1) if softmax else logits
    return logits, out


def forward(x, sd, pool_kernel=(2, 7, 7), softmax=True, stride_mod_layers="", last_stride=1,
            endpoints=None):
    feat = features(x, sd, stride_mod_layers, last_stride, endpoints)
    return head(feat, sd, pool_kernel, softmax)[1]
