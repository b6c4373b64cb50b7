"""1x1x1 conv timing with the epilogue forms of the backward pass (dev tool)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
import torch, ivf_lib as L
lib = L.lib()
B, T, H, W, K, N = 64, 8, 28, 28, 288, 256
if len(sys.argv) > 1 and sys.argv[1] == '4c':
    B, T, H, W, K, N = 64, 4, 14, 14, 296, 512
x = torch.randn(B, T, H, W, K, device='cuda')
w = torch.randn(N, K, 1, 1, 1, device='cuda') * 0.05
wf = torch.empty(lib.ivf_conv3d_pack_fwd_elems(N, K, 1, 1, 1, 1), device='cuda')
L.check(lib.ivf_conv3d_pack_fwd(L.ptr(w), L.ptr(wf), N, K, K, 1, 1, 1, 1, L.stream()))
y = torch.randn(B, T, H, W, N, device='cuda')
gate = torch.randn(B, T, H, W, N, device='cuda')
d = L.ConvDesc()
d.B, d.Ti, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_coff = B, T, H, W, K, K, 0
d.To, d.Ho, d.Wo, d.Cout, d.out_ld, d.out_coff = T, H, W, N, N, 0
d.kT = d.kH = d.kW = 1; d.sT = d.sH = d.sW = 1; d.math = 1
d.mask_ld, d.mask_coff = N, 0
for variant in (2, 7, 8):
    for acc, msk in ((0, 0), (1, 0), (0, 1), (1, 1)):
        d.variant, d.accumulate = variant, acc
        def run(n):
            for _ in range(n):
                L.check(lib.ivf_conv3d(ctypes.byref(d), L.ptr(x), L.ptr(wf), None, None, L.ptr(gate) if msk else None, L.ptr(y), L.stream()))
        run(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(10); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byts = 4.0 * B * T * H * W * (K + N * (1 + acc + msk))
        print(f"variant {variant} acc={acc} mask={msk}: {ms*1e3:7.1f} us  {byts/ms/1e9:5.2f} TB/s  {2.0*B*T*H*W*K*N/ms/1e9:6.1f} TF")
