import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from meta_interpolation_amd import synthetic
from tests.helpers import build_plugin, oracle_base
from oracle import models as OM
torch.set_num_threads(16)
model='voxelflow'
frames = synthetic.septuplet_batch(1, 64, 64, model=model)
base = oracle_base(model)
net = build_plugin(model, 'cuda')
names=[n for n,p in net.named_parameters()]
# forward at theta
f0,f1,t = frames[0],frames[4],frames[2]
out_o = OM.voxelflow_forward(f0,f1,base,{n:base[n] for n in names})
loss_o = torch.nn.functional.mse_loss(out_o,t)
go = torch.autograd.grad(loss_o,[base[n] for n in names],allow_unused=True)
fast = {n:p for n,p in net.named_parameters()}
out_g = net(f0.cuda(),f1.cuda(),params=fast)
loss_g = torch.nn.functional.mse_loss(out_g,t.cuda())
gg = torch.autograd.grad(loss_g,list(fast.values()),allow_unused=True)
print('fwd max abs diff', (out_g.cpu()-out_o).abs().max().item(), 'loss', loss_o.item(), loss_g.item())
for n,a,b in zip(names,go,gg):
    if a is None: continue
    b=b.cpu(); d=(a-b).abs()
    q=np.quantile(a.abs().numpy().ravel(),[0.01,0.1,0.5,0.9])
    flips=((a*b)<0).float().mean().item()
    print('%-22s |g| q1/10/50/90 %s  maxdiff %.2e  rel %.2e  flips %.4f nz %.3f'%(n, np.array2string(q,precision=2), d.max().item(), d.max().item()/a.abs().max().item(), flips, (a!=0).float().mean().item()))

print('---- one Meta-SGD Adamax step from own grads, lr 1e-4')
from oracle import rules as R
lr=1e-4
wo={n:base[n] for n in names}
lrs=R.init_lrs('metasgd', wo, lr)
st=R.RuleState()
with torch.no_grad():
    new_o=R.update_params('metasgd','Adamax',wo,dict(zip(names,go)),lrs,0,st)
    new_g=R.update_params('metasgd','Adamax',{n:p.detach().cpu() for n,p in fast.items()},dict(zip(names,[g.cpu() for g in gg])),lrs,0,R.RuleState())
tot=0;bad=0
for n in names:
    d=(new_o[n]-new_g[n]).abs()
    nb=(d>1e-5).sum().item(); tot+=d.numel(); bad+=nb
    if nb: 
        a=dict(zip(names,go))[n]; b=dict(zip(names,gg))[n].cpu()
        idx=(d>1e-5).nonzero()[:3]
        ex=[(a[tuple(i)].item(), b[tuple(i)].item()) for i in idx]
        print('%-20s changed>1e-5: %d / %d  examples (g_cpu,g_gpu): %s'%(n,nb,d.numel(),ex))
print('total',bad,'/',tot)
# second forward with each side's updated weights, both evaluated on CPU oracle to isolate weight effect
fo={n:new_o[n] for n in names if n in new_o}
fg={n:new_g[n] for n in names if n in new_g}
with torch.no_grad():
    o1=OM.voxelflow_forward(f0,f1,base,fo); o2=OM.voxelflow_forward(f0,f1,base,fg)
print('second fwd (CPU both) mean abs diff due to weight diffs:', (o1-o2).abs().mean().item(), 'max', (o1-o2).abs().max().item())
