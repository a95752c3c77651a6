"""Deviation of every outer-gradient fingerprint from the reference fixture (sepconv_msl_learnable_2step), lockstep and sequential."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import golden, build_system, parse_case_args, fp as helpers_fp
from meta_interpolation_amd import synthetic
name = 'sepconv_msl_learnable_2step'
g = golden("system_" + name)
model = str(g['model'])
for tb in (2, 0):
    system = build_system(model, dict(parse_case_args(g), task_batch=tb))
    system.net.lockstep_tasks = True
    rec = {}
    system.optimizer.step = lambda *a, **k: rec.update({n: helpers_fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
    frames = synthetic.septuplet_batch(2, int(g['H']), int(g['W']), model=model)
    system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
    torch.cuda.synchronize()
    rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
    devs = []
    for k, row in rows.items():
        if abs(row[1]) > 0 and k in rec:
            devs.append((abs(rec[k][0] - row[0]) / max(abs(row[1]), 1e-12), str(k)))
    devs.sort(reverse=True)
    print("task_batch", tb, "H,W", int(g['H']), int(g['W']), "worst:", [(round(d, 6), k[-40:]) for d, k in devs[:6]])
