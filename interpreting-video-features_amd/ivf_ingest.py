"""Clip ingest on the device.

The reference decodes `frame%02d.jpg` files with PIL and converts the stacked uint8 frames
to a float32 [3,T,H,W] tensor on the host (data_loader_jpg.py:23-41,
data_loader_kth.py:20-43); the model then receives 4 bytes per sample over PCIe.  Here the
decoded uint8 frames are uploaded as they are and `ivf_clip_ingest_u8` does the cast and
the permute on the GPU (bit-exact: every uint8 is representable).
"""
import os

import numpy as np
import torch

import ivf_lib as L

NCTHW, CHANNELS_LAST = 0, 1


def decode_clip_u8(folder, clip_size, pattern="frame{:02d}.jpg"):
    """Host side of ImLoader/KTHImLoader.__getitem__: decode `clip_size` frames of a clip
    folder (frames are numbered from 1) -> uint8 [T,H,W,3]."""
    from PIL import Image
    imgs = []
    for i in range(clip_size):
        im = Image.open(os.path.join(folder, pattern.format(i + 1)))
        arr = np.frombuffer(im.tobytes(), dtype=np.uint8)
        imgs.append(arr.reshape((im.size[1], im.size[0], 3)))
    return np.array(imgs)


def ingest_u8(frames, layout=NCTHW, cpad=4, out=None):
    """uint8 frames [T,H,W,C] or [B,T,H,W,C] (numpy, CPU or CUDA tensor) -> float32 device
    tensor [B,C,T,H,W] (NCTHW) or [B,T,H,W,cpad] (CHANNELS_LAST).  An unbatched input gives
    an unbatched result, like the reference's __getitem__."""
    L.require_gpu()
    t = torch.as_tensor(frames)
    if t.dtype != torch.uint8:
        raise TypeError("ingest_u8 expects uint8 frames")
    unbatched = t.dim() == 4
    if unbatched:
        t = t[None]
    if t.dim() != 5:
        raise ValueError("frames must be [T,H,W,C] or [B,T,H,W,C]")
    if not t.is_cuda:
        t = t.contiguous().pin_memory().cuda(non_blocking=True)
    t = t.contiguous()
    B, T, H, W, C = t.shape
    shape = (B, C, T, H, W) if layout == NCTHW else (B, T, H, W, cpad)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=t.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous float32 tensor of shape {shape}")
    L.check(L.lib().ivf_clip_ingest_u8(L.ptr(t), L.ptr(out), B, T, H, W, C, layout, cpad, L.stream()))
    return out[0] if unbatched else out


class JpegFolderLoader:
    """The reference's `DataLoader(ImLoader | KTHImLoader, batch_size, shuffle=False, drop_last=True)`
    as the drivers build it (smth:66-77, KTH:66-84), with the cast + permute on the device:
    yields (sequence float32 [B,3,T,H,W] on the GPU, label LongTensor [B], ids list).

    layout "smth": root/<class id>/<clip id>/frameNN.jpg  (PicDatabase, data_parser.py:121-131;
                   clips in os.walk order, label = class directory name, id = clip directory name)
    layout "kth":  root/<index>/frameNN.jpg + class.txt + label.txt, index = 0 .. len(listdir)-1
                   (data_loader_kth.py:20-47; id = the text of label.txt)
    """

    def __init__(self, root, clip_size=16, batch_size=16, layout="smth", drop_last=True, device=None):
        if layout not in ("smth", "kth"):
            raise ValueError("layout must be 'smth' or 'kth'")
        if not os.path.isdir(root):
            raise FileNotFoundError(f"clip folder '{root}' does not exist")
        self.root, self.clip_size, self.batch_size = root, int(clip_size), int(batch_size)
        self.layout, self.drop_last, self.device = layout, drop_last, device
        self.items = []                      # (folder, label or None, id or None)
        if layout == "smth":
            for cls in next(os.walk(root))[1]:
                for clip in next(os.walk(os.path.join(root, cls)))[1]:
                    self.items.append((os.path.join(root, cls, clip), int(cls), clip))
        else:
            for index in range(len(os.listdir(root))):
                self.items.append((os.path.join(root, str(index)), None, None))

    def __len__(self):
        n = len(self.items)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _item(self, i):
        folder, label, cid = self.items[i]
        frames = decode_clip_u8(folder, self.clip_size)
        if self.layout == "kth":
            with open(os.path.join(folder, "class.txt")) as f:
                label = int(f.readline())
            with open(os.path.join(folder, "label.txt")) as f:
                cid = f.readline()
        return frames, label, cid

    def __iter__(self):
        n = len(self.items)
        stop = n - n % self.batch_size if self.drop_last else n
        for s in range(0, stop, self.batch_size):
            got = [self._item(i) for i in range(s, min(s + self.batch_size, stop))]
            frames = np.stack([g[0] for g in got])
            with torch.cuda.device(self.device if self.device is not None else torch.cuda.current_device()):
                seq = ingest_u8(frames, NCTHW)
            yield seq, torch.tensor([g[1] for g in got], dtype=torch.long), [g[2] for g in got]
