"""Clip ingest kernel timing (dev tool): uint8 [B,16,224,224,3] -> fp32, both layouts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import torch, ivf_ingest
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
u8 = torch.randint(0, 256, (B, 16, 224, 224, 3), dtype=torch.uint8, device='cuda')
for layout, cpad, name in ((ivf_ingest.NCTHW, 4, 'NCTHW'), (ivf_ingest.CHANNELS_LAST, 4, 'channels-last x4')):
    out = ivf_ingest.ingest_u8(u8, layout=layout, cpad=cpad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ivf_ingest.ingest_u8(u8, layout=layout, cpad=cpad, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byts = u8.numel() + out.numel() * 4
    print(f"{name:18s} B={B}: {ms*1e3:7.1f} us  {byts/ms/1e9:5.2f} TB/s (read u8 + write f32 = {byts/1e6:.0f} MB)")
