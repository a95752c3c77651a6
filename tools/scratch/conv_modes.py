import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
mode = sys.argv[1]
if mode == 'benchmark':
    torch.backends.cudnn.benchmark = True
from meta_interpolation_amd import synthetic
from meta_interpolation_amd.config import default_args
from meta_interpolation_amd.meta_learning_system import MODEL_REGISTRY, SceneAdaptiveInterpolation
args = default_args(model='sepconv', num_gpu=1, batch_size=2, number_of_training_steps_per_iter=3, optimizer='SGD')
net = MODEL_REGISTRY['sepconv'](args, False); synthetic.load_seeded_weights(net, 'sepconv')
system = SceneAdaptiveInterpolation(args, net=net.cuda())
frames = [f.cuda() for f in synthetic.septuplet_batch(2, 256, 448)]
t0=time.time(); system.run_train_iter(frames, 0); torch.cuda.synchronize(); print(mode,'first iter %.1f s'%(time.time()-t0))
system.run_train_iter(frames, 0); torch.cuda.synchronize()
t0=time.time()
for _ in range(3): system.run_train_iter(frames, 0)
torch.cuda.synchronize(); dt=(time.time()-t0)/3
print(mode, 'ms/iter %.1f  -> %.1f inner steps/s'%(dt*1e3, 6/dt))
