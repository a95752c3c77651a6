import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_interpolation_amd import synthetic
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
H, W = int(sys.argv[1]), int(sys.argv[2])
res = {}
snaps = {}
for gil in ((1,) if os.environ.get("ONLY_GRAPH") else (0, 1)):
    args = default_args(model='cain', num_gpu=1, batch_size=1, number_of_training_steps_per_iter=1, number_of_evaluation_steps_per_iter=1,
                        optimizer='SGD', loss='1*L1', inner_lr=1e-5, attenuate=not os.environ.get('NO_ATT'), graph_inner_loop=gil)
    net = MODEL_REGISTRY['cain'](args, False)
    synthetic.load_seeded_weights(net, 'cain')
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    if args.attenuate: sd, gm = synthetic.seeded_attenuator_state(len(system.inner_loop_optimizer.names_learning_rates_dict))
    if args.attenuate: system.attenuator.load_state_dict(sd)
    with torch.no_grad():
        if args.attenuate: system.gamma_mult.copy_(gm)
    frames = [f.cuda() for f in synthetic.septuplet_batch(1, H, W, model='cain')]
    grads = {}
    snap = {}
    real = system.optimizer.step
    for it in range(3):
        system.optimizer.step = (lambda *a, **k: (grads.__setitem__(it, {n: p.grad.detach().clone() for n, p in system.named_parameters() if p.grad is not None}), real())[1])
        losses, preds, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
        torch.cuda.synchronize()
        snap[it] = {n: q.detach().clone() for n, q in system.named_parameters()}
        p = preds[0]
        print("gil %d it %d loss %.9g pred mean %.6f std %.6f finite %s ngrads %d" % (gil, it, float(losses['loss']), p.mean().item(), p.std().item(), bool(torch.isfinite(p).all()), len(grads[it])), flush=True)
    res[gil] = grads
    snaps[gil] = snap
for it in range(3 if 0 in res else 0):
    worst = ("", 0.0)
    for k, v in res[0][it].items():
        if k not in res[1][it]:
            print("missing in graph:", k); continue
        d = (res[1][it][k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30)
        if d > worst[1]: worst = (k, d)
    print("it", it, "worst outer-grad rel diff graph vs eager:", worst)
    bad = [(k, (res[1][it][k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30)) for k, v in res[0][it].items() if k in res[1][it]]
    bad = [(k, d) for k, d in bad if not (d < 1e-3)]
    print("   tensors off by > 1e-3:", len(bad), "of", len(res[0][it]), [(k[-50:], float('%.3g' % d)) for k, d in bad[:12]])

for it in range(3 if 0 in snaps else 0):
    rows = []
    for k, v in snaps[0][it].items():
        d = (snaps[1][it][k] - v).abs().max().item()
        rows.append((d, k, v.abs().max().item()))
    rows.sort(reverse=True)
    print("after it", it, "largest parameter differences graph vs eager:", [(round(d, 8), k[-60:], round(m, 5)) for d, k, m in rows[:4]])
    for key in ('gamma_mult',):
        if key in snaps[0][it]:
            print("   gamma_mult eager %r graph %r" % (snaps[0][it][key].item(), snaps[1][it][key].item()))
for it in range(3 if 0 in res else 0):
    for k in ('gamma_mult', 'attenuator.0.weight', 'attenuator.2.bias'):
        if k in res[0][it] and k in res[1][it]:
            print("grad it", it, k, "eager absmax %.4g graph absmax %.4g" % (res[0][it][k].abs().max().item(), res[1][it][k].abs().max().item()))
