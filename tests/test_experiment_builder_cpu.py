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
    system.args.synthetic, system.args.dataset = False, 'middlebury'
    with pytest.raises(NotImplementedError):
        MetaLearningSystemDataLoader(system.args)
