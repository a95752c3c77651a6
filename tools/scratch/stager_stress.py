import os, sys, random, tempfile, types, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from meta_interpolation_amd import data, synthetic
root = synthetic.write_fake_vimeo(tempfile.mkdtemp(), n_train=6, n_test=2)
mk = lambda gpus: types.SimpleNamespace(data_root=root, batch_size=2, val_batch_size=1, test_batch_size=1, mode='train', model='sepconv',
                                        num_gpu=gpus, num_workers=3, random_seed=5, dataset='vimeo90k', synthetic=False)
random.seed(3); cpu = [b for b in data.MetaLearningSystemDataLoader(mk(0)).get_train_batches()]
random.seed(3)
prov = data.MetaLearningSystemDataLoader(mk(1))
big = torch.randn(8192, 8192, device='cuda')
kept = []
for images, meta in prov.get_train_batches():
    # heavy async work on the compute stream + lots of allocator churn, no host sync
    for _ in range(6):
        tmp = big @ big
        tmp2 = torch.empty_like(tmp).normal_()
        del tmp, tmp2
    kept.append([t * 1.0 for t in images])       # consume on the compute stream
    del images
torch.cuda.synchronize()
bad = 0
for (ic, _), ig in zip(cpu, kept):
    for a, b in zip(ic, ig):
        if not torch.equal(a, b.cpu()):
            bad += 1
print('batches', len(kept), 'corrupted tensors', bad)
