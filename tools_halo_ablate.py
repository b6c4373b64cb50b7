import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import torch, ivf_lib as L
lib = L.lib()
# Mixed_3c.b1b-like: B=32 clips, [8,28,28], Cin=128 -> Cout=192, 3x3x3
B, T, H, W, cin, cout = int(os.environ.get('IVF_B', '32')), 8, 28, 28, 128, 192
if len(sys.argv) > 1 and sys.argv[1] == '4f':
    B, T, H, W, cin, cout = int(os.environ.get('IVF_B', '32')), 4, 14, 14, 160, 320
if len(sys.argv) > 1 and sys.argv[1] == '2c':
    B, T, H, W, cin, cout = int(os.environ.get('IVF_B', '32')), 8, 56, 56, 64, 192
x = torch.randn(B, T, H, W, cin, device='cuda')
w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, cin, 3, 3, 3, 1), device='cuda')
L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, cin, cin, 3, 3, 3, 1, L.stream()))
y = torch.empty(B, T, H, W, cout, device='cuda')
d = L.ConvDesc()
d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_coff = B, T, H, W, cin, cin, 0
d.To, d.Ho, d.Wo, d.Cout, d.out_ld, d.out_coff = T, H, W, cout, cout, 0
d.kT = d.kH = d.kW = 3; d.sT = d.sH = d.sW = 1; d.pT = d.pH = d.pW = 1; d.relu = 1; d.math = 1
d.variant = int(os.environ.get("IVF_VARIANT", "0"))
if len(sys.argv) > 1 and sys.argv[1] == 'stembwd':
    # stem backward-data as the plan issues it: 4x4x4 block conv over dY [B,8,112,112,64] -> 32 columns
    B, T, H, W, cin, cout = 32, 8, 112, 112, 64, 32
    x = torch.randn(B, T, H, W, cin, device='cuda')
    wf = torch.randn(cout * 64 * cin, device='cuda') * 0.05      # any values: timing only
    y = torch.empty(B, 16, 224, 224, 4, device='cuda')
    d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_coff = B, T, H, W, cin, cin, 0
    d.To, d.Ho, d.Wo, d.Cout, d.out_ld, d.out_coff = T, H, W, cout, 4, 0
    d.kT = d.kH = d.kW = 4; d.pT = d.pH = d.pW = 2; d.relu = 0
    d.d2s = 1; d.dT, d.dH, d.dW, d.dC = 16, 224, 224, 4; d.bsT = d.bsH = d.bsW = 2
if len(sys.argv) > 1 and sys.argv[1] == 'stemfwd':
    B, T, H, W, cin, cout = 32, 16, 224, 224, 4, 64
    x = torch.randn(B, T, H, W, cin, device='cuda')
    w = torch.randn(cout, 3, 7, 7, 7, device='cuda') * 0.05
    wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(cout, 4, 7, 7, 7, 1), device='cuda')
    L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), cout, 3, 4, 7, 7, 7, 1, L.stream()))
    y = torch.empty(B, 8, 112, 112, cout, device='cuda')
    d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_coff = B, T, H, W, cin, cin, 0
    d.To, d.Ho, d.Wo, d.Cout, d.out_ld, d.out_coff = 8, 112, 112, cout, cout, 0
    d.kT = d.kH = d.kW = 7; d.sT = d.sH = d.sW = 2; d.pT = d.pH = d.pW = 2; d.relu = 1
def run(n):
    for _ in range(n):
        L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), None, None, None, L.ptr(y), L.stream()))
run(3); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(20); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * B * T * H * W * cout * 27 * cin
if len(sys.argv) > 1 and sys.argv[1] in ('stembwd', 'stemfwd'):
    fl = 2.0 * B * 8 * 112 * 112 * 64 * 343 * 3
print(f"var={os.environ.get('IVF_VARIANT','0'):>3} dbg={os.environ.get('IVF_DBG','0'):>3} halo={'off' if os.environ.get('IVF_NO_HALO') else 'on '}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
