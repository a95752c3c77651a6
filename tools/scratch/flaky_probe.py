"""Run-to-run spread of sensitive quantities, with / without the Winograd convolution."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from meta_interpolation_amd import hip_ops, synthetic
from helpers import golden, build_system, fp
sys.path.insert(0, os.path.join(R, 'tests'))
import test_system_gpu as T

for wino in (True, False):
    hip_ops.WINOGRAD_CONV = wino
    for name, key in [('sepconv_msl_learnable_2step', 'inner_loop_optimizer.names_learning_rates_dict.moduleConv4-2-weight'),
                      ('voxelflow_lslr_sgd_2step', None)]:
        g = golden('system_' + name)
        model = str(g['model'])
        vals, losses_ = [], []
        for rep in range(6):
            system = build_system(model, T.parse_case_args(g))
            rec = {}
            system.optimizer.step = lambda *a, **k: rec.update(
                {n: fp(p.grad) for n, p in system.named_parameters() if p.requires_grad and p.grad is not None})
            frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
            losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
            torch.cuda.synchronize()
            losses_.append(losses['loss'].item())
            if key:
                vals.append(rec[key][1])
        want = float(g['train_loss'])
        print('wino', wino, name, 'loss rel dev from fixture:', ['%.2e' % ((l - want) / want) for l in losses_])
        if key:
            rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
            print('   lr grad abs-sum:', ['%.6f' % v for v in vals], 'fixture', rows[key][1])
