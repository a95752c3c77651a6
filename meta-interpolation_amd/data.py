"""Data providers consumed by ExperimentBuilder (reference: data/__init__.py:520-625):
``provider(args=args, current_iter=n)`` with ``.dataset.data_length[split]`` and
``.get_train_batches / .get_val_batches / .get_test_batches(total_batches=...)`` yielding
``(images: list of Tensor[B,3,H,W], metadata: {'imgpaths': ...})``.

* ``VimeoSeptuplet`` / ``HD`` / ``Video``: the reference's readers (data/vimeo_septuplet.py:10-88,
  data/hd_dataset.py:11-79, data/video.py:9-60) —
  same attributes, same ``__getitem__`` results (CPU fp32 CHW tensors) for the same ``random`` state.  Decoding uses
  PIL (the reference's cv2.imread + BGR->RGB swap gives the same RGB bytes for 8-bit PNGs).
* ``FrameStager``: the MI355X feeding path.  Worker threads decode into pinned uint8 HWC buffers (cropped on the
  host, a quarter of the fp32 bytes), one async H2D copy per meta-batch on a side stream, and
  ``savfi_frames_u8_to_f32`` turns the bytes into the 7 (or 4) fp32 ``[B,3,H,W]`` tensors on the GPU while the previous
  meta-iteration still computes.  Bit-identical to the CPU path (same fp32 operation order).
* ``SyntheticSeptupletLoader``: seeded synthetic septuplets (``--synthetic``; bench.py and the parity fixtures).

Middlebury / DAVIS / SNU-FILM readers: out of scope.
"""
import glob
import os
import random
import types
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import synthetic


def _read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'))          # uint8 [H,W,3], RGB  (== cv2.imread(path)[:, :, ::-1])


def _normalisation(model):
    """(div, mean per channel, std) of dst[c] = (u8[c] / div - mean[c]) / std."""
    if model == 'voxelflow':                          # .float() then Normalize(127.5, 127.5)   (:39-41, :74, :80)
        return 1.0, (0.5 * 255,) * 3, 0.5 * 255
    if model == 'superslomo':                         # .float() / 255 then Normalize((.429, .431, .397), 1)   (:31-35)
        return 255.0, (0.429, 0.431, 0.397), 1.0
    return 255.0, (0.0,) * 3, 1.0                     # .float() / 255                         (:76)


def _to_float_chw(u8_hwc, model):
    div, mean, std = _normalisation(model)
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(u8_hwc, (2, 0, 1)))).float()
    if div != 1.0:
        t = t / div
    if any(m != 0.0 for m in mean) or std != 1.0:
        t = (t - torch.tensor(mean, dtype=t.dtype).view(3, 1, 1)) / std
    return t


class VimeoSeptuplet(object):
    """data/vimeo_septuplet.py:10-88.  `sequences/<a>/<b>/im{1..7}.png`, `sep_trainlist.txt`, `sep_testlist.txt`."""

    def __init__(self, args):
        self.args = args
        self.data_root = args.data_root
        self.image_root = os.path.join(self.data_root, 'sequences')
        with open(os.path.join(self.data_root, 'sep_trainlist.txt'), 'r') as f:
            self.trainlist = f.read().splitlines()
        with open(os.path.join(self.data_root, 'sep_testlist.txt'), 'r') as f:
            self.testlist = f.read().splitlines()
        self.batch_size = {'train': args.batch_size, 'val': args.val_batch_size, 'test': args.test_batch_size}
        self.crop_size = 256
        self.frames = [1, 2, 3, 4, 5, 6, 7]
        self.current_set_name = "train" if args.mode == 'train' else 'val'
        self.data_length = {'train': len(self.trainlist), 'val': len(self.testlist), 'test': 0}

    def plan(self, index):
        """Paths + crop + flip of item `index`.  Consumes `random` exactly like the reference (randint(h), randint(w),
        random() per training item) and only reads the first frame's header, so that plans can be drawn in item order
        while the files are decoded in parallel."""
        lst = self.trainlist if self.current_set_name == 'train' else self.testlist
        imgpath = os.path.join(self.image_root, lst[index % len(lst)])
        imgpaths = ['%s/im%d.png' % (imgpath, i) for i in self.frames]
        crop = None
        if self.current_set_name == 'train':
            from PIL import Image
            with Image.open(imgpaths[0]) as im:
                W, H = im.size
            rnd_h = random.randint(0, max(0, H - self.crop_size))
            rnd_w = random.randint(0, max(0, W - self.crop_size))
            crop = (rnd_h, rnd_w)
            if random.random() >= 0.5:
                imgpaths = imgpaths[::-1]
        return imgpaths, crop

    def load(self, plan):
        """uint8 RGB frames [7,H,W,3] of a plan."""
        imgpaths, crop = plan
        images = [_read_rgb(p) for p in imgpaths]
        if crop is not None:
            y, x = crop
            images = [v[y:y + self.crop_size, x:x + self.crop_size, :] for v in images]
        return np.stack(images), imgpaths

    def decode(self, index):
        return self.load(self.plan(index))

    def __getitem__(self, index):
        frames, imgpaths = self.decode(index)
        return [_to_float_chw(f, self.args.model) for f in frames], {'imgpaths': imgpaths}

    def switch_set(self, set_name, current_iter=None):
        self.current_set_name = set_name

    def __len__(self):
        return self.data_length[self.current_set_name]


class Video(object):
    """data/video.py:9-60: sliding 4-frame clips over `data_root/*.<img_fmt>` for `--mode test` (ToTensor = /255)."""

    def __init__(self, args):
        self.args = args
        images = sorted(glob.glob(os.path.join(args.data_root, '*.%s' % args.img_fmt)))
        for im in images:
            try:
                float(im.split('_')[-1][:-4])
            except ValueError:                        # the reference renames frames in place to "<name>_0.000000.<fmt>"
                os.rename(im, '%s_%.06f.%s' % (im[:-4], 0.0, args.img_fmt))
        images = sorted(glob.glob(os.path.join(args.data_root, '*.%s' % args.img_fmt)))
        if len(images) < 4:
            print("Not enough frames for fast adaptation!")
            images = images + [images[-1]] * (4 - len(images))
            self.imglist = [images]
        else:
            self.imglist = [[images[i], images[i + 1], images[i + 2], images[i + 3]] for i in range(len(images) - 3)]
        print('[%d] images ready to be loaded' % len(self.imglist))
        self.batch_size = {'train': 0, 'val': 0, 'test': args.test_batch_size}
        self.data_length = {'train': 0, 'val': 0, 'test': len(self.imglist)}
        self.current_set_name = 'test'

    def plan(self, index):
        return list(self.imglist[index]), None

    def load(self, plan):
        return np.stack([_read_rgb(p) for p in plan[0]]), plan[0]

    def decode(self, index):
        return self.load(self.plan(index))

    def __getitem__(self, index):
        frames, imgpaths = self.decode(index)
        return [_to_float_chw(f, 'other') for f in frames], {'imgpaths': imgpaths}      # ToTensor: always /255

    def switch_set(self, set_name, current_iter=None):
        self.current_set_name = set_name

    def __len__(self):
        return self.data_length[self.current_set_name]


class HD(object):
    """data/hd_dataset.py:11-79: every video directory under `data_root`, 7-frame windows with stride 2 (the last
    windows are clamped to the final 7 frames; shorter videos give one short window), validation split only."""

    def __init__(self, args):
        self.args = args
        self.data_root = args.data_root
        self.image_root = self.data_root
        vidlist = sorted(glob.glob(os.path.join(self.image_root, '*')))
        imglist = [sorted(glob.glob(os.path.join(v, '*.png'))) for v in vidlist]
        n_frames = 7
        self.imgBatch = []
        for frames in imglist:
            t = 0
            while t < len(frames):
                if len(frames) >= n_frames:
                    self.imgBatch.append(frames[t:t + n_frames] if t + n_frames <= len(frames) else frames[-n_frames:])
                else:
                    self.imgBatch.append(frames)
                t += 2
        self.batch_size = {'train': 1, 'val': 1, 'test': 1}
        self.current_set_name = 'val'
        self.data_length = {'train': 0, 'val': len(self.imgBatch), 'test': 0}
        _normalisation(args.model)

    def plan(self, index):
        return list(self.imgBatch[index]), None

    def load(self, plan):
        return np.stack([_read_rgb(p) for p in plan[0]]), plan[0]

    def decode(self, index):
        return self.load(self.plan(index))

    def __getitem__(self, index):
        frames, imgpaths = self.decode(index)
        return [_to_float_chw(f, self.args.model) for f in frames], {'imgpaths': imgpaths}

    def switch_set(self, set_name, current_iter=None):
        self.current_set_name = set_name

    def __len__(self):
        return self.data_length[self.current_set_name]


def _to_float_chw_totensor(u8_hwc, model):
    """The tensor pipeline of the two evaluation-only readers below: transforms.ToTensor() (uint8 / 255) FIRST, then the plugin's
    normalisation on it -- VoxelFlow: Normalize(127.5, 127.5)(t * 255), which is not bit-identical to normalising the uint8 values
    directly (data/middlebury.py:88-93, data/snufilm.py:53-56 vs data/vimeo_septuplet.py:74-80)."""
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(u8_hwc, (2, 0, 1)))).float().div(255)
    if model == 'voxelflow':
        t = (t * 255.0 - torch.full((3, 1, 1), 0.5 * 255)) / torch.full((3, 1, 1), 0.5 * 255)
    elif model == 'superslomo':
        t = (t - torch.tensor((0.429, 0.431, 0.397)).view(3, 1, 1)) / torch.ones(3, 1, 1)
    return t


class Middlebury(object):
    """data/middlebury.py:12-105 (`--dataset middlebury`): the 'other' scenes with eight input frames; frames 09, 10, 11, 12 are the
    inputs, `other-gt-interp/<scene>/frame10i11.png` the target, returned as a septuplet-shaped list [f0, 0, f1, gt, f2, 0, f3]
    (zeros where Vimeo has frames this set does not).  Validation split only."""

    def __init__(self, args):
        self.args = args
        self.data_root = args.data_root
        self.image_root = os.path.join(self.data_root, 'other-data-all')
        self.gt_root = os.path.join(self.data_root, 'other-gt-interp')
        self.imglist, self.gt_list = [], []
        for d in sorted(glob.glob(self.image_root + '/*')):
            frames = sorted(glob.glob(d + '/*.png'))
            if len(frames) == 8:                       # scenes with two frames only are skipped, other counts are ignored (:37-42)
                self.imglist.append(frames[2:6])
                self.gt_list.append(os.path.join(self.gt_root, d.split('/')[-1], 'frame10i11.png'))
        self.batch_size = {'train': 1, 'val': 1, 'test': 1}
        self.current_set_name = 'val'
        self.data_length = {'train': 0, 'val': len(self.imglist), 'test': 0}

    def __getitem__(self, index):
        paths, gt_path = self.imglist[index], self.gt_list[index]
        imgs = [_to_float_chw_totensor(_read_rgb(p), self.args.model) for p in paths]
        gt = _to_float_chw_totensor(_read_rgb(gt_path), self.args.model)
        dummy = torch.zeros_like(gt)
        return ([imgs[0], dummy, imgs[1], gt, imgs[2], dummy, imgs[3]],
                {'imgpaths': [paths[0], "", paths[1], gt_path, paths[2], "", paths[3]]})

    def switch_set(self, set_name, current_iter=None):
        self.current_set_name = set_name

    def __len__(self):
        return self.data_length[self.current_set_name]


class SNUFILM(object):
    """data/snufilm.py:8-69 (`--dataset snufilm`): `data_root/test-hard-meta.txt`, one quintuplet of image paths per line, returned
    as [f0, 0, f1, f2, f3, 0, f4].  Validation split only."""

    def __init__(self, args):
        self.args = args
        with open(os.path.join(args.data_root, 'test-hard-meta.txt'), 'r') as f:
            self.frame_list = [v.split(' ') for v in f.read().splitlines()]
        self.batch_size = {'train': 1, 'val': 1, 'test': 1}
        self.current_set_name = 'val'
        self.data_length = {'train': 0, 'val': len(self.frame_list), 'test': 0}
        print("Test dataset has %d quintuplets" % len(self.frame_list))

    def __getitem__(self, index):
        paths = self.frame_list[index]
        imgs = [_to_float_chw_totensor(_read_rgb(p), self.args.model) for p in paths]
        dummy = torch.zeros_like(imgs[0])
        return (imgs[:1] + [dummy] + imgs[1:4] + [dummy] + imgs[-1:],
                {'imgpaths': paths[:1] + [''] + paths[1:4] + [''] + paths[-1:]})

    def switch_set(self, set_name, current_iter=None):
        self.current_set_name = set_name

    def __len__(self):
        return self.data_length[self.current_set_name]


class FrameStager(object):
    """Decoded uint8 frames -> fp32 [B,3,H,W] tensors on the GPU (see the module docstring).

    `stage(items)` takes B decoded items (each uint8 [F,H,W,3]); returns a handle whose `.tensors()` gives the F tensors
    after making the current stream wait for the side stream.  Two pinned / device byte buffers alternate, so batch
    k+1 can be staged while batch k is being consumed."""

    def __init__(self, device, model):
        from . import _hip
        self._hip = _hip
        self.device = torch.device(device)
        self.div, self.mean, self.std = _normalisation(model)
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = [None, None]
        self._turn = 0

    def stage(self, items):
        frames = np.stack(items, axis=1) if not isinstance(items, np.ndarray) else items      # [F,B,H,W,3]
        F, B, H, W, _ = frames.shape
        slot = self._slots[self._turn]
        if slot is None or slot[0].shape != frames.shape:
            if slot is not None:
                slot[2].synchronize()                               # its last copy / kernel may still be in flight
            with torch.cuda.stream(self.stream):
                # device bytes from the SIDE stream's pool (see `out` below): a block the compute stream just freed may
                # still be in use by kernels queued there, and the H2D copy on this stream would not wait for them
                dev_u8 = torch.empty(frames.shape, dtype=torch.uint8, device=self.device)
            slot = (torch.empty(frames.shape, dtype=torch.uint8).pin_memory(), dev_u8, torch.cuda.Event())
            self._slots[self._turn] = slot
        self._turn ^= 1
        pinned, dev_u8, done = slot
        done.synchronize()                                          # the copy that last used this pinned buffer finished
        pinned.numpy()[...] = frames
        with torch.cuda.stream(self.stream):
            # allocated from the SIDE stream's pool: a block the compute stream has just freed may still be read or
            # written by kernels queued there, and this stream does not wait for them
            out = torch.empty((F, B, 3, H, W), dtype=torch.float32, device=self.device)
            dev_u8.copy_(pinned, non_blocking=True)
            lib = self._hip.lib()
            self._hip.check(lib.savfi_frames_u8_to_f32(dev_u8.data_ptr(), out.data_ptr(), F * B, H, W, 0, self.div, *self.mean,
                                                       self.std, self.stream.cuda_stream), "savfi_frames_u8_to_f32")
            done.record(self.stream)
        ready = torch.cuda.Event()
        ready.record(self.stream)
        return _Staged(out, ready)


class _Staged(object):
    def __init__(self, out, ready):
        self._out, self._ready = out, ready

    def tensors(self):
        cur = torch.cuda.current_stream()
        cur.wait_event(self._ready)
        self._out.record_stream(cur)        # consumed on the compute stream: the side-stream pool must not recycle it early
        return [self._out[f] for f in range(self._out.shape[0])]


class DatasetProvider(object):
    """MetaLearningSystemDataLoader over a reader (data/__init__.py:520-625).  Batches are assembled by `num_workers`
    decode threads; on a GPU they go through FrameStager one batch ahead of the consumer, otherwise (CPU tensors) through
    the reader's own `__getitem__`."""

    def __init__(self, args, dataset, current_iter=0, task_parallel=None):
        from .task_parallel import TaskParallel
        self.args = args
        self.dataset = dataset
        # one process per GPU: rank r adapts tasks {t : t mod G == r} of every meta-batch and only decodes those; the
        # other items of the batch stay zero (never read), their `random` draws are still consumed so that crops and
        # flips do not depend on the number of ranks
        self.task_parallel = task_parallel if task_parallel is not None else TaskParallel()
        self.batch_size = {'train': args.batch_size, 'val': args.val_batch_size, 'test': args.test_batch_size}
        self.num_workers = max(1, int(getattr(args, 'num_workers', 1)))
        self.full_data_length = dict(dataset.data_length)
        self.total_train_iters_produced = current_iter * self.batch_size['train']
        use_gpu = torch.cuda.is_available() and getattr(args, 'num_gpu', 0) > 0
        norm_model = args.model if isinstance(dataset, (VimeoSeptuplet, HD)) else 'other'          # Video: ToTensor, always /255
        self.stager = FrameStager(torch.device('cuda', torch.cuda.current_device()), norm_model) if use_gpu else None
        self._shuffle = torch.Generator().manual_seed(int(getattr(args, 'random_seed', 0)))

    def _index_batches(self, mode):
        n = self.dataset.data_length[mode]
        order = torch.randperm(n, generator=self._shuffle).tolist() if mode == 'train' else list(range(n))
        bs = max(1, self.batch_size[mode])
        return [order[i:i + bs] for i in range(0, n, bs)]            # drop_last=False, like the reference's DataLoader

    def _decode_batch(self, pool, idxs, shard):
        plans = [self.dataset.plan(i) for i in idxs]          # `random` draws in item order (reproducible)
        mine = set(self.task_parallel.local_tasks(len(idxs))) if shard else set(range(len(idxs)))
        order = sorted(mine)
        loaded = dict(zip(order, pool.map(self.dataset.load, [plans[t] for t in order])))   # PNG decode in parallel
        ref = loaded[order[0]][0] if order else self.dataset.load(plans[0])[0]
        frames = [loaded[t][0] if t in loaded else np.zeros_like(ref) for t in range(len(idxs))]
        paths = [[plans[t][0][f] for t in range(len(idxs))] for f in range(len(plans[0][0]))]
        return frames, {'imgpaths': paths}

    def _batches(self, mode):
        batches = self._index_batches(mode)
        shard = mode == 'train' and self.task_parallel.active      # validation / test sweeps run on every rank
        with ThreadPoolExecutor(self.num_workers) as pool:
            if self.stager is None:
                for idxs in batches:
                    frames, meta = self._decode_batch(pool, idxs, shard)
                    model = self.args.model if isinstance(self.dataset, (VimeoSeptuplet, HD)) else 'other'
                    yield [torch.stack([_to_float_chw(fr[f], model) for fr in frames]) for f in range(frames[0].shape[0])], meta
                return
            pending = None
            for idxs in batches:                                   # one batch staged ahead of the one being consumed
                frames, meta = self._decode_batch(pool, idxs, shard)
                staged = (self.stager.stage(frames), meta)
                if pending is not None:
                    yield pending[0].tensors(), pending[1]
                pending = staged
            if pending is not None:
                yield pending[0].tensors(), pending[1]

    def _limit(self, mode, total_batches):
        if total_batches == -1:
            self.dataset.data_length = dict(self.full_data_length)
        else:
            self.dataset.data_length[mode] = total_batches * self.dataset.batch_size[mode]

    def get_train_batches(self, total_batches=-1, augment_images=False):
        self._limit('train', total_batches)
        self.dataset.switch_set(set_name="train", current_iter=self.total_train_iters_produced)
        self.total_train_iters_produced += self.batch_size["train"]
        return self._batches('train')

    def get_val_batches(self, total_batches=-1, augment_images=False):
        self._limit('val', total_batches)
        self.dataset.switch_set(set_name="val")
        return self._batches('val')

    def get_test_batches(self, total_batches=-1, augment_images=False):
        self._limit('test', total_batches)
        self.dataset.switch_set(set_name='test')
        return self._batches('test')


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
    """String dispatch of data/__init__.py:537-556 for the datasets this build covers."""
    if getattr(args, 'synthetic', False):
        return SyntheticSeptupletLoader(args, current_iter)
    if args.dataset == 'vimeo90k':
        return DatasetProvider(args, VimeoSeptuplet(args), current_iter)
    if args.dataset == 'hd':
        return DatasetProvider(args, HD(args), current_iter)
    if args.dataset == 'test':
        return DatasetProvider(args, Video(args), current_iter)
    if args.dataset == 'middlebury':
        return DatasetProvider(args, Middlebury(args), current_iter)
    if args.dataset == 'snufilm':
        return DatasetProvider(args, SNUFILM(args), current_iter)
    # 'davis': the reference's dispatch names data/davis.py, which is not in its repository (data/__init__.py:546-548)
    raise NotImplementedError("dataset %r: the reference ships no reader for it (vimeo90k, hd, middlebury, snufilm, test, or --synthetic)" % args.dataset)
