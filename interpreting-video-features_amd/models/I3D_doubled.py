"""Drop-in for video_features_pytorch/models/I3D_doubled.py: same `Model`
constructor, state_dict keys and call convention; executed as a hand-written HIP
plan on the MI355X (no torch.nn compute)."""
from models._i3d_module import (I3DBase, InceptionModule, MaxPool3dSamePadding,  # noqa: F401
                                Unit3D)


class Model(I3DBase):
    """Inception-v1 I3D (I3D_doubled.py:149-388); head AvgPool3d [2*..,7,7]."""
    _HEAD_HW = (7, 7)

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint='Logits', name='inception_i3d',
                 in_channels=3, dropout_keep_prob=0.5, last_stride=1, stride_mod_layers=[], softMax=False,
                 lastRelu=None):
        super().__init__()
        self._construct(num_classes, spatial_squeeze, final_endpoint, name, in_channels, dropout_keep_prob,
                        last_stride, stride_mod_layers, softMax, lastRelu, head_time_base=2)
