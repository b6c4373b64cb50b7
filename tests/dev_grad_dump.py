import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ivf_engine, ivf_recipe as R, ivf_arch as arch
sd_np = R.i3d_state_dict(num_classes=174)
x = torch.from_numpy(R.clip(7))[None]
eng = ivf_engine.I3DEngine(174, (3,16,224,224), max_batch=1)
eng.load_state_dict(sd_np)
p = eng.forward(x.cuda())
t = int(p[0].argmax())
score, dx = eng.backward(1, target=[t])
out = {'dx': dx.cpu().numpy(), 'probs': p.cpu().numpy()}
for n in ('Mixed_5b','MaxPool3d_5a_2x2','Mixed_4f','Mixed_4e','Mixed_4d','Mixed_4c', 'Mixed_4b', 'MaxPool3d_4a_3x3'):
    out[n+':grad'] = eng.endpoint(n+':grad',1).cpu().numpy()
    out[n] = eng.endpoint(n,1).cpu().numpy()
np.savez_compressed(os.path.join(ROOT,'gpurun_out','grads.npz'), **out)
