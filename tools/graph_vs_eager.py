"""hipGraph replays of the inner loop against the eager loop on CHANGING frames, any plugin and frame size:

    python tools/graph_vs_eager.py MODEL H W [--iters 3] [--steps 1] [--batch 1] [--l2f]

Three meta-iterations with real outer steps on different synthetic septuplets and an eager allocation between them; prints the
losses of both modes and the worst relative deviation of an outer gradient per iteration (the per-process summation-order noise of
the eager loop itself is 1e-6 .. 1e-3 depending on the plugin: DESIGN.md 7).  Written after the stale captured mean at 1280x720
(DESIGN.md 9.6): a capture that depends on state a replay does not restore shows up here as a deviation that grows with the
iteration, or only from the second replay on.
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('model'); ap.add_argument('H', type=int); ap.add_argument('W', type=int)
ap.add_argument('--iters', type=int, default=3); ap.add_argument('--steps', type=int, default=1)
ap.add_argument('--batch', type=int, default=1); ap.add_argument('--l2f', action='store_true')
ap.add_argument('--no-step', action='store_true', help='skip the outer optimizer step (weights stay at their seeded values)')
ap.add_argument('--perturb', type=float, default=0.0, help='relative perturbation of the SECOND run\'s frames (sensitivity control)')
ap.add_argument('--modes', default='0,1', help="graph_inner_loop of the two runs compared ('0,0': the eager loop against itself)")
opt = ap.parse_args()
import torch
from meta_interpolation_amd import synthetic
if opt.l2f:
    from meta_interpolation_amd import graph_inner_loop
    graph_inner_loop.GRAPH_L2F = True
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
runs = {}
modes = [int(v) for v in opt.modes.split(',')]
for slot, gil in enumerate(modes):
    args = default_args(model=opt.model, num_gpu=1, batch_size=opt.batch, number_of_training_steps_per_iter=opt.steps,
                        number_of_evaluation_steps_per_iter=opt.steps, optimizer='SGD', loss='1*L1', inner_lr=1e-5,
                        attenuate=bool(opt.l2f), graph_inner_loop=gil, task_batch=0, lazy_logging=0)
    net = MODEL_REGISTRY[opt.model](args, False)
    synthetic.load_seeded_weights(net, opt.model)
    system = SceneAdaptiveInterpolation(args, net=net.cuda())
    if opt.l2f:
        sd, gm = synthetic.seeded_attenuator_state(len(system.inner_loop_optimizer.names_learning_rates_dict))
        system.attenuator.load_state_dict(sd)
        with torch.no_grad():
            system.gamma_mult.copy_(gm)
    seen, losses = [], []
    step = system.optimizer.step
    system.optimizer.step = lambda *a, **k: (seen.append({n: p.grad.detach().clone() for n, p in system.named_parameters()
                                                          if p.grad is not None}), None if opt.no_step else step(*a, **k))[1]
    for it in range(opt.iters):
        frames = [f.cuda() for f in synthetic.septuplet_batch(opt.batch, opt.H, opt.W, model=opt.model, first_task=it * opt.batch)]
        if slot == 1 and opt.perturb:
            frames = [f * (1.0 + opt.perturb) for f in frames]
        out, _, _ = system.run_train_iter(data_batch=frames, epoch=0, do_evaluation=False)
        losses.append(float(out['loss']))
        filler = torch.full((1 << 26,), 7.0, device='cuda'); del filler
    runs[slot] = (losses, seen, len(getattr(system, '_graphs', {})))
    del system, net
    torch.cuda.empty_cache()
print("%s %dx%d batch %d steps %d l2f %d modes %s: graph sets captured %d" % (opt.model, opt.H, opt.W, opt.batch, opt.steps, opt.l2f, opt.modes, runs[1][2]))
for it in range(opt.iters):
    e, g = runs[0][1][it], runs[1][1][it]
    worst = max(((g[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30), k) for k, v in e.items() if k in g)
    print("  it %d loss first %.9g second %.9g | worst outer-gradient deviation %.3e (%s); tensors %d / %d" % (
        it, runs[0][0][it], runs[1][0][it], worst[0], worst[1][-48:], len(g), len(e)), flush=True)
    rows = sorted(((g[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30), k, v.abs().max().item()) for k, v in e.items() if k in g)[::-1][:6]
    print("      " + "  ".join("%s %.1e (|g| %.1e)" % (k[-28:], d, m) for d, k, m in rows), flush=True)
