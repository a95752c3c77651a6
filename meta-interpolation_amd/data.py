"""Data-provider surface consumed by ExperimentBuilder (reference: data/__init__.py:520-625):
``provider(args=args, current_iter=n)`` with ``.dataset.data_length[split]`` and
``.get_train_batches / .get_val_batches / .get_test_batches(total_batches=...)`` yielding
``(images: list of Tensor[B,3,H,W], metadata: {'imgpaths': ...})``.

Only the seeded synthetic provider lives here (``--synthetic``): the PNG dataset readers
(VimeoSeptuplet, HD, Video, ...) are host-side decode and out of scope for this path.
"""
import types

from . import synthetic


class SyntheticSeptupletLoader(object):
    """Deterministic stand-in for MetaLearningSystemDataLoader: task t of split s is septuplet(seed(s)+t)."""

    def __init__(self, args, current_iter=0, height=256, width=448, length=None):
        self.args = args
        self.height, self.width = height, width
        n = length or {'train': 64, 'val': 8, 'test': 8}
        self.dataset = types.SimpleNamespace(data_length=dict(n))
        self.current_iter = current_iter
        self.model = 'voxelflow' if args.model == 'voxelflow' else 'other'

    def _batches(self, split, batch_size, total_batches, frames, offset):
        for b in range(total_batches):
            first = offset + (b * batch_size) % max(self.dataset.data_length[split], 1)
            images = synthetic.septuplet_batch(batch_size, self.height, self.width, model=self.model,
                                               first_task=first, frames=frames)
            paths = [['synthetic/%s/%05d/im%d.png' % (split, first + t, f + 1) for t in range(batch_size)]
                     for f in range(frames)]
            yield images, {'imgpaths': paths}

    def get_train_batches(self, total_batches=-1):
        total = self.dataset.data_length['train'] // self.args.batch_size if total_batches < 0 else total_batches
        return self._batches('train', self.args.batch_size, total, 7, 0)

    def get_val_batches(self, total_batches=-1):
        total = self.dataset.data_length['val'] // self.args.val_batch_size if total_batches < 0 else total_batches
        return self._batches('val', self.args.val_batch_size, total, 7, 100000)

    def get_test_batches(self, total_batches=-1):
        total = self.dataset.data_length['test'] // self.args.test_batch_size if total_batches < 0 else total_batches
        return self._batches('test', self.args.test_batch_size, total, 4, 200000)


def MetaLearningSystemDataLoader(args, current_iter=0):
    if not getattr(args, 'synthetic', False):
        raise NotImplementedError(
            "dataset readers are outside the inner-loop path of this build; run with --synthetic "
            "or pass your own provider with the MetaLearningSystemDataLoader surface")
    return SyntheticSeptupletLoader(args, current_iter)
