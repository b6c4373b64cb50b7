"""Drop-in for video_features_pytorch/models/I3D_doubled_kth.py (120x160 KTH
clips: head AvgPool3d [finalTimeLength,4,5], I3D_doubled_kth.py:302-308)."""
from models._i3d_module import (I3DBase, InceptionModule, MaxPool3dSamePadding,  # noqa: F401
                                Unit3D)


class Model(I3DBase):
    _HEAD_HW = (4, 5)

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint='Logits', name='inception_i3d',
                 in_channels=3, dropout_keep_prob=0.5, last_stride=1, stride_mod_layers=[], finalTimeLength=2,
                 softMax=False, lastRelu=None):
        super().__init__()
        self.finalTimeLength = finalTimeLength
        self._construct(num_classes, spatial_squeeze, final_endpoint, name, in_channels, dropout_keep_prob,
                        last_stride, stride_mod_layers, softMax, lastRelu, head_time_base=finalTimeLength)
