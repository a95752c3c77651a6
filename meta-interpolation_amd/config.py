"""Command-line flags of the experiment entry point -- same names and defaults as the reference's
config.py:14-77 so that its launch scripts (scripts/run_*.sh) keep working.  ``get_args()`` returns
``(args, unparsed)`` and derives ``args.cuda`` from ``--num_gpu`` (reference :79-89).

Additions for the MI355X build (all optional, all default to the reference behaviour):
  --fuse_support_pairs {0,1}   run the two support triplets of an inner step as one N=2 forward
  --fuse_conv_act {0,1}        conv+bias+(Leaky)ReLU with fused epilogue kernels (default 1; first-order only)
  --graph_inner_loop {-1,0,1}  replay the first-order inner loop from captured hipGraphs (graph_inner_loop.py): 1 always, 0 never,
                               -1 (default) when the rank would otherwise adapt its tasks one at a time (launch-bound passes)
  --sepconv_window {0,1}       SepConv: evaluate the sub-networks / 51-tap op on the frame window only (same values)
  --wgrad_overlap {0,1}        weight gradients of first-order support passes on a side stream, beside the data-gradient chain (default 0)
  --task_streams N             adapt N tasks of a meta-batch concurrently (one Python thread + HIP stream each); default -1:
                               1 in the eager loops, up to 4 where single tasks are replayed from hipGraphs
  --task_batch T               adapt up to T tasks of a meta-batch in LOCKSTEP: one launch per layer for all of them, per-task fast
                               weights (default 8; first order -- second order, L2F on partly routed plugins and T <= 1 take the
                               reference's sequential task loop)
  --synthetic                  feed seeded synthetic septuplets instead of reading a dataset
"""
import argparse

# (flag, type | 'flag', default)  grouped like the reference
_FLAGS = {
    'Dataset': [
        ('dataset', str, 'vimeo90k'), ('num_frames', int, 3), ('data_root', str, 'data/vimeo_septuplet'),
        ('img_fmt', str, 'png'), ('fps', int, 30),
    ],
    'Model': [
        ('model', str, 'CAIN'), ('depth', int, 3), ('n_resblocks', int, 12), ('up_mode', str, 'shuffle'),
    ],
    'Learning': [
        ('mode', str, 'train'), ('loss', str, '1*L1'), ('optimizer', str, 'Adam'),
        ('inner_lr', float, 1e-5), ('outer_lr', float, 1e-5), ('beta1', float, 0.9), ('beta2', float, 0.99),
        ('weight_decay', float, 1e-4), ('batch_size', int, 8), ('val_batch_size', int, 1),
        ('test_batch_size', int, 1), ('test_mode', str, 'hard'), ('start_epoch', int, 0),
        ('max_epoch', int, 60), ('resume', 'flag', False), ('resume_exp', str, None),
        ('pretrained_model', str, None), ('fix_loaded', 'flag', False),
        ('number_of_training_steps_per_iter', int, 1), ('number_of_evaluation_steps_per_iter', int, 1),
        ('learnable_per_layer_per_step_inner_loop_learning_rate', 'flag', False),
        ('enable_inner_loop_optimizable_bn_params', 'flag', False), ('second_order', 'flag', False),
        ('first_order_to_second_order_epoch', int, -1), ('use_multi_step_loss_optimization', 'flag', False),
        ('multi_step_loss_num_epochs', int, 1), ('total_iter_per_epoch', int, 10),
        ('attenuate', 'flag', False), ('metasgd', 'flag', False),
    ],
    'Misc': [
        ('exp_name', str, 'exp'), ('log_iter', int, 20), ('log_dir', str, 'logs'), ('eval_iter', int, 10),
        ('data_dir', str, 'data'), ('num_gpu', int, 1), ('random_seed', int, 12345), ('num_workers', int, 5),
        ('use_tensorboard', 'flag', False), ('viz', 'flag', False), ('lpips', 'flag', False),
    ],
    'MI355X': [
        ('fuse_support_pairs', int, 1), ('fuse_conv_act', int, 1), ('graph_inner_loop', int, -1), ('sepconv_window', int, 1),
        ('task_streams', int, -1), ('wgrad_overlap', int, 0), ('task_batch', int, 8), ('lazy_logging', int, 1),
        ('synthetic', 'flag', False),
    ],
}

_CHOICES = {'mode': ['train', 'val', 'test']}


def build_parser():
    parser = argparse.ArgumentParser(description='scene-adaptive video frame interpolation (MI355X build)')
    for group_name, flags in _FLAGS.items():
        group = parser.add_argument_group(group_name)
        for name, kind, default in flags:
            if kind == 'flag':
                group.add_argument('--' + name, action='store_true')
            else:
                group.add_argument('--' + name, type=kind, default=default, choices=_CHOICES.get(name))
    return parser


def finalize(args):
    args.cuda = args.num_gpu > 0
    return args


def get_args(argv=None):
    args, unparsed = build_parser().parse_known_args(argv)
    finalize(args)
    if len(unparsed) > 1:
        print("Unparsed args: {}".format(unparsed))
    return args, unparsed


def default_args(**overrides):
    """Namespace with every default, for programmatic use (tests, bench)."""
    args, _ = get_args([])
    for k, v in overrides.items():
        if not hasattr(args, k):
            raise AttributeError("unknown flag '%s'" % k)
        setattr(args, k, v)
    return finalize(args)
