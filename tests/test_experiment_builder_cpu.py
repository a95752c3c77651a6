"""CPU: the ExperimentBuilder entry-point surface (train loop -> validation sweep -> checkpoint ->
scheduler; val and test modes; HD half-split) driven with the toy CPU plugin and synthetic data."""
import os

import torch

from meta_interpolation_amd import utils
from meta_interpolation_amd.data import MetaLearningSystemDataLoader, SyntheticSeptupletLoader
from meta_interpolation_amd.experiment_builder import ExperimentBuilder
from tests.helpers import build_toy_system


def _provider(h, w, length):
    return lambda args, current_iter=0: SyntheticSeptupletLoader(args, current_iter, height=h, width=w, length=length)


def test_train_loop_validates_checkpoints_and_resumes(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    system = build_toy_system(batch=2, steps=1)
    args = system.args
    args.synthetic, args.total_iter_per_epoch, args.max_epoch, args.exp_name, args.log_iter = True, 2, 1, 'toy', 2
    before = {k: v.clone() for k, v in system.state_dict().items()}
    eb = ExperimentBuilder(args, _provider(16, 24, {'train': 8, 'val': 2, 'test': 2}), system)
    log = eb.run_experiment()
    assert eb.state['current_iter'] == 2 and eb.epoch == 1 and len(log) >= 1
    ckpt = torch.load(os.path.join('checkpoint', 'toy', 'checkpoint.pth'), weights_only=False)
    assert set(ckpt) == {'epoch', 'arch', 'state_dict', 'best_PSNR'} and ckpt['epoch'] == 1
    assert os.path.exists(os.path.join('checkpoint', 'toy', 'model_best.pth'))
    assert any(not torch.equal(before[k], v) for k, v in system.state_dict().items())      # it trained
    # resume: a fresh system picks the weights up by name and shape
    fresh = build_toy_system(batch=2, steps=1, seed=123)
    fresh.args.exp_name, fresh.args.resume_exp = 'toy', None
    utils.load_checkpoint(fresh.args, fresh, None)
    assert fresh.args.start_epoch == 1
    for k, v in system.state_dict().items():
        assert torch.equal(fresh.state_dict()[k], v), k


def test_val_and_test_modes_and_hd_split(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    system = build_toy_system(batch=1, steps=1)
    args = system.args
    args.synthetic = True
    args.mode = 'val'
    # 720x704 > 5e5 pixels -> evaluated as two halves and stitched (reference experiment_builder.py:105-115)
    eb = ExperimentBuilder(args, _provider(720, 704, {'train': 1, 'val': 1, 'test': 1}), system)
    calls = []
    orig = system.run_validation_iter
    system.run_validation_iter = lambda data_batch: (calls.append(tuple(data_batch[0].shape)), orig(data_batch))[1]
    losses, acc = eb.run_experiment()
    assert calls == [(1, 3, 360, 704), (1, 3, 360, 704)] and acc['psnr'].count == 1 and 'total' in losses
    args.mode = 'test'
    eb = ExperimentBuilder(args, _provider(16, 24, {'train': 1, 'val': 1, 'test': 2}), system)
    outs = eb.run_experiment()
    assert len(outs) == 2 and outs[0][0].shape == (3, 16, 24)


def test_train_loop_runs_from_a_vimeo_directory_with_the_toy_plugin(tmp_path, monkeypatch):
    """main.py's wiring: MetaLearningSystemDataLoader(args) -> VimeoSeptuplet reader -> ExperimentBuilder (CPU tensors)."""
    from meta_interpolation_amd import synthetic
    monkeypatch.chdir(tmp_path)
    system = build_toy_system(batch=2, steps=1)
    args = system.args
    args.synthetic, args.dataset, args.data_root = False, 'vimeo90k', synthetic.write_fake_vimeo(str(tmp_path / 'vimeo'))
    args.total_iter_per_epoch, args.max_epoch, args.exp_name, args.log_iter, args.num_workers = 2, 1, 'toyv', 2, 2
    eb = ExperimentBuilder(args, MetaLearningSystemDataLoader, system)
    eb.run_experiment()
    assert eb.state['current_iter'] == 2 and eb.epoch == 1
    assert os.path.exists(os.path.join('checkpoint', 'toyv', 'checkpoint.pth'))


def test_unsupported_datasets_fail_loudly():
    import pytest
    system = build_toy_system(batch=1, steps=1)
    system.args.synthetic, system.args.dataset = False, 'davis'          # the one name of the reference's dispatch without a reader in its repository
    with pytest.raises(NotImplementedError):
        MetaLearningSystemDataLoader(system.args)


def test_val_and_test_modes_write_frames_like_the_reference(tmp_path, monkeypatch):
    """experiment_builder.py:194-206 (test: `<prefix>_<mid time stamp>.png` next to the clip) and :228-234 (val:
    checkpoint/<exp>/<dataset>/<a>/<b>/im4.png), through the real readers."""
    from PIL import Image
    import numpy as np
    from meta_interpolation_amd import synthetic
    monkeypatch.chdir(tmp_path)
    system = build_toy_system(batch=1, steps=1)
    args = system.args
    args.synthetic, args.exp_name, args.num_workers, args.img_fmt = False, 'writer', 2, 'png'
    # val mode over a vimeo directory
    args.mode, args.dataset, args.data_root = 'val', 'vimeo90k', synthetic.write_fake_vimeo(str(tmp_path / 'vimeo'), height=32, width=48)
    ExperimentBuilder(args, MetaLearningSystemDataLoader, system).run_experiment()
    written = sorted(os.path.relpath(os.path.join(r, f), 'checkpoint/writer/vimeo90k')
                     for r, _, fs in os.walk('checkpoint/writer/vimeo90k') for f in fs)
    assert written == ['00002/0001/im4.png', '00002/0002/im4.png']
    assert np.asarray(Image.open('checkpoint/writer/vimeo90k/00002/0001/im4.png')).shape == (32, 48, 3)
    # test mode over a folder of frames
    args.mode, args.dataset, args.data_root = 'test', 'test', synthetic.write_fake_video(str(tmp_path / 'clip'), n_frames=5, height=16, width=24)
    outs = ExperimentBuilder(args, MetaLearningSystemDataLoader, system).run_experiment()
    assert len(outs) == 2
    new = sorted(f for f in os.listdir(args.data_root) if not f.endswith('_0.000000.png'))
    assert new == ['frame_0.500000.png']        # both clips map to the same name: stamps 0 / 0 -> (0 + 1.0) / 2, prefix 'frame'


def test_load_checkpoint_follows_the_reference_rules(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    a = build_toy_system(batch=1, steps=1)
    utils.save_checkpoint({'epoch': 7, 'state_dict': a.state_dict(), 'optimizer': a.optimizer.state_dict()}, True, 'src')
    b = build_toy_system(batch=1, steps=1, seed=5)
    b.args.exp_name, b.args.resume_exp, b.args.mode = 'dst', 'src', 'train'
    utils.load_checkpoint(b.args, b, b.optimizer)
    assert b.args.start_epoch == 0                                   # another experiment's weights: restart the epochs
    assert all(torch.equal(v, a.state_dict()[k]) for k, v in b.state_dict().items())
    b.args.exp_name = 'src'
    utils.load_checkpoint(b.args, b, b.optimizer)
    assert b.args.start_epoch == 7
    b.args.mode = 'val'                                               # val / test read model_best.pth
    utils.load_checkpoint(b.args, b, None)
