"""Run every reference-generated system fixture through the HIP product path and print the deviations
(one JSON line per case and phase).  The -m gpu tests assert on these quantities; this tool shows the
margins.  Usage (GPU box): python tools/parity_report.py [--deterministic 0|1] [--repeat N] > gpurun_out/parity.jsonl

--deterministic 1 (default) pins MIOpen to its deterministic solvers like tests/conftest.py does; with 0 the report
shows one draw of MIOpen's run-to-run noise on top."""
import argparse
import json
import os
import sys
import tempfile

os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='savfi_parity_miopen_')   # see bench.py

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from meta_interpolation_amd import synthetic  # noqa: E402
from tests.helpers import build_system, golden, observe, parse_case_args  # noqa: E402

CASES = ['c1_cain_lslr_sgd', 'cain_l2f', 'cain_lslr_adam_1step', 'sepconv_lslr_sgd_2step',
         'sepconv_metasgd_adamax_2step', 'sepconv_msl_learnable_2step', 'voxelflow_metasgd_adamax_2step',
         'voxelflow_lslr_sgd_2step', 'voxelflow_script_metasgd_adam_1step', 'rrin_lslr_sgd_2step', 'superslomo_lslr_sgd_2step']


def fp_dev(got, want):
    scale = max(abs(want[1]), 1e-12)
    return max(abs(got[0] - want[0]), abs(got[1] - want[1])) / scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--deterministic', type=int, default=1)
    ap.add_argument('--repeat', type=int, default=1)
    ap.add_argument('--cases', default=','.join(CASES))
    o = ap.parse_args()
    torch.backends.cudnn.deterministic = bool(o.deterministic)
    for name in [c for c in o.cases.split(',') for _ in range(o.repeat)]:
        g = golden("system_" + name)
        model = str(g['model'])
        for phase in ('train', 'val'):
            system = build_system(model, dict(parse_case_args(g), task_batch=0))
            rec = observe(system, check_rule=True)
            frames = synthetic.septuplet_batch(int(g['B']), int(g['H']), int(g['W']), model=model)
            if phase == 'train':
                losses, preds, metrics = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=True)
            else:
                losses, preds, metrics = system.run_validation_iter(data_batch=frames)
            torch.cuda.synchronize()
            got = torch.stack([p.squeeze(0) for p in preds]).cpu().numpy()
            row = dict(case=name, phase=phase, miopen_deterministic=bool(o.deterministic),
                       loss_rel=abs(losses['loss'].item() - float(g[phase + '_loss'])) / abs(float(g[phase + '_loss'])),
                       pixel_l1=float(np.abs(got - g[phase + '_preds']).mean()),
                       dpsnr=abs(metrics['psnr'].avg - float(g[phase + '_psnr'])),
                       dssim=abs(float(metrics['ssim'].avg) - float(g[phase + '_ssim'])),
                       n_live=rec['n_live'], rule_err_vs_oracle=max(rec['rule_err']) if rec['rule_err'] else None)
            wdev = gdev = 0.0
            for i, d in enumerate(rec['weight_fp']):
                for k, r in zip(list(g['%s_weight_fp_%d_keys' % (phase, i)]), g['%s_weight_fp_%d' % (phase, i)]):
                    wdev = max(wdev, fp_dev(d[k], r))
            for i, d in enumerate(rec['grad_fp']):
                for k, r in zip(list(g['%s_grad_fp_%d_keys' % (phase, i)]), g['%s_grad_fp_%d' % (phase, i)]):
                    gdev = max(gdev, fp_dev(d[k], r))
            row['weight_fp_dev'], row['grad_fp_dev'] = wdev, gdev
            if phase == 'train':
                rows = dict(zip(list(g['outer_grad_fp_0_keys']), g['outer_grad_fp_0']))
                row['outer_keys_equal'] = set(rows) == set(rec['outer_grad_fp'])
                row['outer_fp_dev'] = max(fp_dev(rec['outer_grad_fp'][k], r) for k, r in rows.items()
                                          if k in rec['outer_grad_fp'])
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
