import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ivf_engine, ivf_recipe as R, ivf_arch as arch
from oracle import i3d_ref
torch.set_num_threads(16)
sd_np = R.i3d_state_dict(num_classes=174)
sd = R.to_torch(sd_np)
x = torch.from_numpy(R.clip(7))[None]
eps = {}
xr = x.clone().requires_grad_()
feat = i3d_ref.features(xr, sd, endpoints=eps)
for v in eps.values(): v.retain_grad()
logits, out = i3d_ref.head(feat, sd)
t = int(out[0].argmax())
out[0, t].backward()
eng = ivf_engine.I3DEngine(174, (3,16,224,224), max_batch=1)
eng.load_state_dict(sd_np)
p = eng.forward(x.cuda())
score, dx = eng.backward(1, target=[t])
def rel(a, b): return float((a-b).abs().max() / b.abs().max())
name = 'Mixed_5c'
X = eps['Mixed_5b'].detach().clone().requires_grad_()
gY = eps['Mixed_5c'].grad
b0 = i3d_ref.unit3d(X, sd, name + '.b0')
t1 = i3d_ref.unit3d(X, sd, name + '.b1a'); t1.retain_grad()
b1 = i3d_ref.unit3d(t1, sd, name + '.b1b')
t2 = i3d_ref.unit3d(X, sd, name + '.b2a'); t2.retain_grad()
b2 = i3d_ref.unit3d(t2, sd, name + '.b2b')
t3 = i3d_ref.maxpool_same(X, (3,3,3), (1,1,1)); t3.retain_grad()
b3 = i3d_ref.unit3d(t3, sd, name + '.b3b')
Y = torch.cat([b0,b1,b2,b3], 1)
print("Y fwd", rel(eng.endpoint(name,1).cpu(), Y.detach()))
for nm, tt in (('.b1a', t1), ('.b2a', t2), ('.b3a', t3)):
    print("fwd", nm, rel(eng.endpoint(name+nm,1).cpu(), tt.detach()))
(Y * gY).sum().backward()
print("grad t1", rel(eng.endpoint(name+'.b1a:grad',1).cpu(), t1.grad * (t1>0).float()))
print("grad t2", rel(eng.endpoint(name+'.b2a:grad',1).cpu(), t2.grad * (t2>0).float()))
print("grad t3", rel(eng.endpoint(name+'.b3a:grad',1).cpu(), t3.grad))
print("grad X ", rel(eng.endpoint('Mixed_5b:grad',1).cpu(), X.grad * (X>0).float()))
# per-branch contributions to grad X
for label, br in (('b0', b0), ('b1', b1), ('b2', b2), ('b3', b3)):
    pass
g = eng.endpoint('Mixed_5b:grad',1).cpu()
ref = X.grad * (X>0).float()
d = (g-ref).abs()
idx = d.flatten().topk(5).indices
for i in idx.tolist():
    c, r = divmod(i, 2*7*7); print("worst", c, r, float(g.flatten()[i]), float(ref.flatten()[i]))
