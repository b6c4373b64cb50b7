import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import ivf_lib as L
if os.environ.get("IVF_DIAG_LIB"): L.LIB_PATH = os.path.join(ROOT, "interpreting-video-features_amd", os.environ["IVF_DIAG_LIB"])
import torch
lib = L.lib()
cases = {'3c.b3a': (64, 8, 28, 28, 480, (3,3,3), (1,1,1)), '2a': (64, 8, 112, 112, 64, (1,3,3), (1,2,2)), '3a': (64, 8, 56, 56, 192, (1,3,3), (1,2,2)), '4a': (64, 8, 28, 28, 480, (3,3,3), (2,2,2)),
         '4f.b3a': (64, 4, 14, 14, 528, (3,3,3), (1,1,1))}
import ivf_arch as arch
for name, (B, T, H, W, C, k, s) in cases.items():
    BF = os.environ.get("IVF_POOL_BF16") == "1"      # bf16 activation storage (PoolDesc.act_bf16)
    dt = torch.bfloat16 if BF else torch.float32
    x = torch.relu(torch.randn(B, T, H, W, C, device='cuda')).to(dt)
    pads = [arch.same_pad(n, kk, ss)[0] for n, kk, ss in zip((T,H,W), k, s)]
    outs = [arch.out_size(n, kk, ss) for n, kk, ss in zip((T,H,W), k, s)]
    y = torch.empty(B, *outs, C, device='cuda', dtype=dt); idx = torch.empty(B, *outs, C, dtype=torch.uint8, device='cuda')
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    esz = 2 if BF else 4
    d = L.PoolDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.C, d.in_ld, d.in_coff = B, T, H, W, C, C, 0
    d.To, d.Ho, d.Wo, d.out_ld, d.out_coff = *outs, C, 0
    d.kT, d.kH, d.kW = k; d.sT, d.sH, d.sW = s; d.pT, d.pH, d.pW = pads
    d.act_bf16 = 1 if BF else 0
    def fwd(): L.check(lib.ivf_maxpool3d_fwd(ctypes.byref(d), L.ptr(x), L.ptr(y), L.ptr(idx), L.stream()))
    def bwd(): L.check(lib.ivf_maxpool3d_bwd(ctypes.byref(d), L.ptr(dy), L.ptr(idx), L.ptr(dx), L.ptr(x), 0, L.stream()))
    for fn, label, bytes_ in ((fwd, 'fwd', x.numel()*esz + y.numel()*(esz+1)), (bwd, 'bwd', dy.numel()*(esz+1) + dx.numel()*2*esz)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:8s} {label} {ms*1e3:8.1f} us  {bytes_/ms/1e9:6.2f} TB/s (compulsory bytes)   no_fixed={os.environ.get('IVF_POOL_NO_FIXED','')}")
