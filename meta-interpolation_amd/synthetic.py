"""Deterministic synthetic inputs: seeded weights for the three plugins and seeded septuplets.

There is no network for datasets or checkpoints, and the reference ships no weights
(pretrained_models/_dummy_model.pth is empty), so benchmarks, parity tests and the golden fixtures
all use THIS recipe (SURVEY.md section 8c "Weights", 8d "Synthetic inputs").  Everything is numpy
RandomState based, so the same bytes come out on the GPU box, in this container, and inside the
reference import that generated tests/golden.
"""
import zlib

import numpy as np
import torch

# Xavier-uniform bound multiplier per plugin: keeps un-normalised random nets at O(1) outputs
# (raw Xavier makes CAIN's 127-conv stack blow up to L1 ~ 20; SepConv's separable taps need 1.2x to leave ~0).
_GAIN = {'sepconv': 1.2, 'cain': 0.5, 'voxelflow': 1.0, 'rrin': 1.0, 'superslomo': 1.68}
# per sub-network overrides (name prefix): RRIN's flow nets need > 1 to produce flows of a sizeable fraction of a pixel,
# its `final` residual net < 1 so that the clamp(0, 1) at the end does not saturate
_GAIN_PREFIX = {'rrin': {'Flow_L.': 1.3, 'refine_flow.': 1.3, 'final.': 0.85}}
# Super SloMo works on mean-subtracted frames (data/vimeo_septuplet.py:31-33; meta_learning_system.py:71-73 undoes it)
SUPERSLOMO_MEAN = (0.429, 0.431, 0.397)


def _stream(seed, name):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7fffffff)


# VoxelFlow conv weights: normal(0, 0.01) is the model's own initialisation (voxelflow/core/models/voxel_flow.py:267-274).  The default
# recipe triples it so that a random-weight network produces flows of a sizeable fraction of a pixel (every code path of the warp is
# exercised) -- which also makes 5 Adamax steps on a noise-like texture chaotic.  Recipe 'smooth' keeps the model's own scale: sub-pixel,
# smooth flows, an iteration that reproduces itself in the reference (tests/golden/full_c3s_*.npz: the testable C3 contract).
_VOXELFLOW_STD_MULT = {None: 3.0, 'default': 3.0, 'smooth': 1.0}


def seeded_state_dict(net, model, seed=12345, recipe=None):
    """{name: tensor} for every parameter and buffer of `net`, one independent numpy stream per name."""
    out = {}
    for name, ref in net.state_dict().items():
        gain = _GAIN.get(model, 1.0)
        for prefix, g in _GAIN_PREFIX.get(model, {}).items():
            if name.startswith(prefix):
                gain = g
        rs = _stream(seed, name)
        shape = tuple(ref.shape)
        if name.endswith('num_batches_tracked'):
            val = np.zeros(shape, dtype=np.int64)
        elif name.endswith('running_mean'):
            val = rs.uniform(-0.1, 0.1, size=shape)
        elif name.endswith('running_var'):
            val = rs.uniform(0.5, 1.5, size=shape)
        elif len(shape) == 4:  # conv weight [out, in, kh, kw]
            fan_in = shape[1] * shape[2] * shape[3]
            fan_out = shape[0] * shape[2] * shape[3]
            if model == 'voxelflow':
                val = rs.normal(0.0, 0.01, size=shape) * _VOXELFLOW_STD_MULT[recipe]
            else:
                bound = gain * np.sqrt(6.0 / (fan_in + fan_out))
                val = rs.uniform(-bound, bound, size=shape)
        elif name.endswith('_bn.weight'):
            val = rs.uniform(0.8, 1.2, size=shape)
        else:  # biases (conv bias, BN bias)
            val = rs.uniform(-0.02, 0.02, size=shape)
        out[name] = torch.from_numpy(np.asarray(val)).to(ref.dtype)
    return out


def load_seeded_weights(net, model, seed=12345, recipe=None):
    sd = seeded_state_dict(net, model, seed, recipe)
    net.load_state_dict(sd)
    return sd


def _box_blur(a, k):
    """k x k box filter (valid) with cumulative sums."""
    c = np.cumsum(np.cumsum(np.pad(a, ((1, 0), (1, 0), (0, 0))), axis=0), axis=1)
    return (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / float(k * k)


def septuplet(task_seed, height, width, frames=7, model='sepconv'):
    """7 frames [3,H,W] float32 of a smooth texture translating by (1, 2) px per frame, quantised to
    k/255 like decoded PNGs.  VoxelFlow gets the (255 x - 127.5)/127.5 normalisation of the
    reference's loader (data/vimeo_septuplet.py:38-40)."""
    rs = np.random.RandomState(1234 + task_seed)
    k = 9
    hh, ww = height + (frames - 1) * 1 + k - 1, width + (frames - 1) * 2 + k - 1
    tex = _box_blur(rs.uniform(0.0, 1.0, size=(hh, ww, 3)), k)
    tex = np.clip((tex - 0.5) * 6.0 + 0.5, 0.0, 1.0)
    out = []
    for f in range(frames):
        crop = tex[f * 1:f * 1 + height, f * 2:f * 2 + width]
        q = np.round(crop * 255.0) / 255.0
        if model == 'voxelflow':
            q = (255.0 * q - 127.5) / 127.5
        elif model == 'superslomo':
            q = q - np.asarray(SUPERSLOMO_MEAN)
        out.append(torch.from_numpy(np.ascontiguousarray(q.transpose(2, 0, 1)).astype(np.float32)))
    return out


def septuplet_batch(num_tasks, height, width, model='sepconv', first_task=0, frames=7):
    """list of `frames` tensors [B,3,H,W] -- the data_batch layout run_train_iter consumes."""
    tasks = [septuplet(first_task + t, height, width, frames, model) for t in range(num_tasks)]
    return [torch.stack([tasks[t][f] for t in range(num_tasks)], 0) for f in range(frames)]


def seeded_attenuator_state(num_layers, seed=777, gamma_mult=0.5):
    """L2F attenuator (Linear(L,L)-ReLU-Linear(L,L)-Sigmoid) + gamma_mult with a non-trivial seeded
    state (the reference initialises gamma_mult to 0, which turns L2F into a no-op)."""
    rs = np.random.RandomState(seed)
    shapes = [('0.weight', (num_layers, num_layers)), ('0.bias', (num_layers,)),
              ('2.weight', (num_layers, num_layers)), ('2.bias', (num_layers,))]
    sd = {k: torch.from_numpy(rs.uniform(-0.05, 0.05, size=s).astype(np.float32)) for k, s in shapes}
    return sd, torch.full((1,), float(gamma_mult))


# --------------------------------------------------------------------------------------------
# tiny on-disk datasets in the reference's layouts (tests, fixtures, smoke runs of the data pipeline)
# --------------------------------------------------------------------------------------------
def _fake_frame(rng, height, width, k):
    base = rng.randint(0, 256, size=(height // 4 + 2, width // 4 + 2, 3)).astype(np.float32)
    img = np.kron(base, np.ones((4, 4, 1), dtype=np.float32))[:height, :width]
    img = np.roll(img, shift=(k, 2 * k), axis=(0, 1)) + rng.randint(0, 8, size=(height, width, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def write_fake_vimeo(root, n_train=3, n_test=2, height=260, width=272, seed=99):
    """`root/sequences/000XX/000Y/im{1..7}.png` + `sep_trainlist.txt` / `sep_testlist.txt` (data/vimeo_septuplet.py:14-21)."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    lists = {'sep_trainlist.txt': [], 'sep_testlist.txt': []}
    for split, n, name in (('train', n_train, 'sep_trainlist.txt'), ('test', n_test, 'sep_testlist.txt')):
        for s in range(n):
            rel = '%05d/%04d' % (1 if split == 'train' else 2, s + 1)
            os.makedirs(os.path.join(root, 'sequences', rel), exist_ok=True)
            for k in range(7):
                Image.fromarray(_fake_frame(rng, height, width, k)).save(os.path.join(root, 'sequences', rel, 'im%d.png' % (k + 1)))
            lists[name].append(rel)
    for name, rows in lists.items():
        with open(os.path.join(root, name), 'w') as f:
            f.write('\n'.join(rows))
    return root


def write_fake_video(root, n_frames=6, height=96, width=128, seed=7):
    """`root/frame_000K_0.000000.png`: the layout data/video.py:13-27 leaves after its in-place renaming."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    for k in range(n_frames):
        Image.fromarray(_fake_frame(rng, height, width, k)).save(os.path.join(root, 'frame_%04d_%.06f.png' % (k, 0.0)))
    return root


def write_fake_hd(root, lengths=(9, 7, 4), height=48, width=64, seed=21):
    """`root/<video>/<frame>.png` for data/hd_dataset.py:18-40 (one long, one exactly-7 and one short video)."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    for v, n in enumerate(lengths):
        os.makedirs(os.path.join(root, 'video%02d' % v), exist_ok=True)
        for k in range(n):
            Image.fromarray(_fake_frame(rng, height, width, k)).save(os.path.join(root, 'video%02d' % v, '%04d.png' % k))
    return root


def write_fake_middlebury(root, height=48, width=64, seed=33):
    """`root/other-data-all/<scene>/frame07..14.png` (8 frames; one scene with only two is skipped by the reader) and
    `root/other-gt-interp/<scene>/frame10i11.png` for data/middlebury.py:26-44."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    for scene, n in (('Beanbags', 8), ('Dimetrodon', 2), ('Walking', 8)):
        os.makedirs(os.path.join(root, 'other-data-all', scene), exist_ok=True)
        os.makedirs(os.path.join(root, 'other-gt-interp', scene), exist_ok=True)
        first = 7 if n == 8 else 10
        for k in range(n):
            Image.fromarray(_fake_frame(rng, height, width, k)).save(os.path.join(root, 'other-data-all', scene, 'frame%02d.png' % (first + k)))
        Image.fromarray(_fake_frame(rng, height, width, 99)).save(os.path.join(root, 'other-gt-interp', scene, 'frame10i11.png'))
    return root


def write_fake_snufilm(root, clips=3, height=48, width=64, seed=44):
    """`root/test-hard-meta.txt`: one line of five space-separated image paths per sample (data/snufilm.py:15-19), frames under root/test/."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    lines = []
    for c in range(clips):
        d = os.path.join(root, 'test', 'clip%02d' % c)
        os.makedirs(d, exist_ok=True)
        paths = []
        for k in range(5):
            path = os.path.join(d, '%05d.png' % (4 * k))
            Image.fromarray(_fake_frame(rng, height, width, k)).save(path)
            paths.append(path)
        lines.append(' '.join(paths))
    with open(os.path.join(root, 'test-hard-meta.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return root

