"""python -m meta_interpolation_amd.main --model sepconv --synthetic ...   (reference: main.py:1-11)

Under torch.distributed.run (one process per GPU) the meta-batch is sharded across ranks with one
RCCL all-reduce of outer gradients per iteration; with a single process it is the sequential loop."""
from .config import get_args
from .data import MetaLearningSystemDataLoader
from .experiment_builder import ExperimentBuilder
from .meta_learning_system import SceneAdaptiveInterpolation
from . import task_parallel


def main(argv=None):
    args, _ = get_args(argv)
    if args.cuda:
        task_parallel.init_from_env()
    print(args)
    model = SceneAdaptiveInterpolation(args)
    data = MetaLearningSystemDataLoader
    savfi_system = ExperimentBuilder(model=model, data=data, args=args)
    return savfi_system.run_experiment()


if __name__ == '__main__':
    main()
