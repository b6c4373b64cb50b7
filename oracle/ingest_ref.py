"""Oracle (test infrastructure): clip ingest restated on numpy.

Follows video_features_pytorch/data_loader_jpg.py:23-41 (ImLoader.__getitem__) and
data_loader_kth.py:20-43 (KTHImLoader.__getitem__): frames decoded with PIL to uint8
[H,W,3], stacked to [T,H,W,3], cast to float32 (exact for 0..255) and permuted to
[3,T,H,W].  Pinned by tests/golden/ingest.npz (the reference loaders run on synthetic
JPEG folders).
"""
import io

import numpy as np


def decode_frames(jpeg_bytes_list):
    """data_loader_jpg.py:26-31: PIL decode, raw bytes reshaped [H, W, 3]."""
    from PIL import Image
    imgs = []
    for b in jpeg_bytes_list:
        im = Image.open(io.BytesIO(bytes(b)))
        arr = np.frombuffer(im.tobytes(), dtype=np.uint8)
        imgs.append(arr.reshape((im.size[1], im.size[0], 3)))
    return np.array(imgs)


def to_model_input(frames_u8):
    """data_loader_jpg.py:32-37: [T,H,W,C] uint8 -> float32 [C,T,H,W]."""
    return np.ascontiguousarray(frames_u8.astype(np.float32).transpose(3, 0, 1, 2))


def to_channels_last(frames_u8, cpad):
    """The plan's input layout: [T,H,W,cpad] float32, pad lanes zero."""
    out = np.zeros(frames_u8.shape[:-1] + (cpad,), dtype=np.float32)
    out[..., :frames_u8.shape[-1]] = frames_u8
    return out
