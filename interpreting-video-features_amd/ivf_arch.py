"""Static description of the Inception-v1 I3D graph used on the saliency path.

This is the single source of truth for layer names, channel tables, kernel
shapes and strides; the HIP network plan (csrc/i3d_net.hip via `_engine.py`),
the deterministic weight recipe (`recipe.py`) and the CPU oracle all walk it.

Reference: video_features_pytorch/models/I3D_doubled.py:165-184 (endpoint
order), :229-307 (channel tables), :311-334 (head); the KTH variant differs
only in the head average-pool window (I3D_doubled_kth.py:302-308).
"""

ENDPOINTS = (
    'Conv3d_1a_7x7', 'MaxPool3d_2a_3x3', 'Conv3d_2b_1x1', 'Conv3d_2c_3x3',
    'MaxPool3d_3a_3x3', 'Mixed_3b', 'Mixed_3c', 'MaxPool3d_4a_3x3',
    'Mixed_4b', 'Mixed_4c', 'Mixed_4d', 'Mixed_4e', 'Mixed_4f',
    'MaxPool3d_5a_2x2', 'Mixed_5b', 'Mixed_5c',
)

# name -> (in_channels, [b0, b1a, b1b, b2a, b2b, b3b])
INCEPTION = {
    'Mixed_3b': (192, [64, 96, 128, 16, 32, 32]),
    'Mixed_3c': (256, [128, 128, 192, 32, 96, 64]),
    'Mixed_4b': (480, [192, 96, 208, 16, 48, 64]),
    'Mixed_4c': (512, [160, 112, 224, 24, 64, 64]),
    'Mixed_4d': (512, [128, 128, 256, 24, 64, 64]),
    'Mixed_4e': (512, [112, 144, 288, 32, 64, 64]),
    'Mixed_4f': (528, [256, 160, 320, 32, 128, 128]),
    'Mixed_5b': (832, [256, 160, 320, 32, 128, 128]),
    'Mixed_5c': (832, [384, 192, 384, 48, 128, 128]),
}

# name -> (kernel, default stride); the temporal stride of the three entries
# marked in STRIDE_MOD can be replaced by `last_stride` (I3D_doubled.py:224-227,
# 262-266, 292-296).
POOLS = {
    'MaxPool3d_2a_3x3': ((1, 3, 3), (1, 2, 2)),
    'MaxPool3d_3a_3x3': ((1, 3, 3), (1, 2, 2)),
    'MaxPool3d_4a_3x3': ((3, 3, 3), (2, 2, 2)),
    'MaxPool3d_5a_2x2': ((2, 2, 2), (2, 2, 2)),
}
STRIDE_MOD = ('Conv3d_1a_7x7', 'MaxPool3d_4a_3x3', 'MaxPool3d_5a_2x2')

FEATURE_CHANNELS = 1024  # Mixed_5c output = 384 + 384 + 128 + 128


def parse_stride_mod(stride_mod_layers):
    """The reference tests membership with `in` on whatever it was given
    (a str gives substring semantics, a list gives element semantics)."""
    if stride_mod_layers is None:
        # I3D_doubled.py:224 raises TypeError on `x in None`; mirror it.
        raise TypeError("argument of type 'NoneType' is not iterable")
    return stride_mod_layers


def temporal_stride(end_point, stride_mod_layers, last_stride):
    return last_stride if end_point in stride_mod_layers else 2


def conv_units(in_channels=3, num_classes=400, stride_mod_layers="", last_stride=1):
    """Yield (state_dict prefix, cin, cout, kernel, stride, has_bn) for every
    Unit3D in registration order (I3D_doubled.py:229-334)."""
    stride_mod_layers = parse_stride_mod(stride_mod_layers)
    st = temporal_stride('Conv3d_1a_7x7', stride_mod_layers, last_stride)
    yield ('Conv3d_1a_7x7', in_channels, 64, (7, 7, 7), (st, 2, 2), True)
    yield ('Conv3d_2b_1x1', 64, 64, (1, 1, 1), (1, 1, 1), True)
    yield ('Conv3d_2c_3x3', 64, 192, (3, 3, 3), (1, 1, 1), True)
    for name in ENDPOINTS:
        if name not in INCEPTION:
            continue
        cin, oc = INCEPTION[name]
        yield (name + '.b0', cin, oc[0], (1, 1, 1), (1, 1, 1), True)
        yield (name + '.b1a', cin, oc[1], (1, 1, 1), (1, 1, 1), True)
        yield (name + '.b1b', oc[1], oc[2], (3, 3, 3), (1, 1, 1), True)
        yield (name + '.b2a', cin, oc[3], (1, 1, 1), (1, 1, 1), True)
        yield (name + '.b2b', oc[3], oc[4], (3, 3, 3), (1, 1, 1), True)
        yield (name + '.b3b', cin, oc[5], (1, 1, 1), (1, 1, 1), True)
    yield ('logits', FEATURE_CHANNELS, num_classes, (1, 1, 1), (1, 1, 1), False)


def head_time_kernel(stride_mod_layers, last_stride, base=2):
    """Temporal extent of the head AvgPool3d (I3D_doubled.py:311-318; `base`
    is `finalTimeLength` in I3D_doubled_kth.py:302-308)."""
    if stride_mod_layers == "" or stride_mod_layers is None:
        return base
    return int(base * ((2 / last_stride) ** len(stride_mod_layers.split(','))))


def same_pad(n, k, s):
    """TF-'same' total padding along one dim (I3D_doubled.py:9-13, 77-81);
    front gets pad//2, back the rest (:29-34, :94-99)."""
    p = max(k - s, 0) if n % s == 0 else max(k - (n % s), 0)
    return p // 2, p - p // 2


def out_size(n, k, s):
    f, b = same_pad(n, k, s)
    return (n + f + b - k) // s + 1
