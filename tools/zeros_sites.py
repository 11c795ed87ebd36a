import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv=['x','--no-cpu-baseline']
import bench
from dose_prediction_amd import _lib, losses, synth
from dose_prediction_amd.optim import FusedAdam
args = bench.parse(); dev = torch.device('cuda:0'); shape=(128,128,128)
net = bench.build_model(args, shape, dev)
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, weight_decay=3e-5, amsgrad=True)
x, gt = synth.dose_input(2, shape).to(dev), synth.dose_target(2, shape).to(dev)
def step():
    opt.zero_grad(set_to_none=True); out = net(x); loss = losses.gen_loss(out, gt, 10.0, 1.0, casecade=True, freez=True); loss.backward(); opt.step()
step(); step()
cnt = collections.Counter()
oz, ozl = torch.zeros, torch.zeros_like
def wrap(f, name):
    def g(*a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        cnt[name + " @ " + " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st))] += 1
        return f(*a, **k)
    return g
torch.zeros, torch.zeros_like = wrap(oz, "zeros"), wrap(ozl, "zeros_like")
oz_ = torch.Tensor.zero_
def zz(self): cnt["zero_ @ " + " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(traceback.extract_stack(limit=4)[:-1]))] += 1; return oz_(self)
torch.Tensor.zero_ = zz
oc_, ocl_ = torch.Tensor.copy_, torch.Tensor.clone
def site(): return " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(traceback.extract_stack(limit=5)[:-2]))
def cc(self, *a, **k): cnt["copy_ @ " + site()] += 1; return oc_(self, *a, **k)
def cl(self, *a, **k): cnt["clone @ " + site()] += 1; return ocl_(self, *a, **k)
torch.Tensor.copy_, torch.Tensor.clone = cc, cl
step(); torch.cuda.synchronize()
for k, v in cnt.most_common(40): print(v, k)
