"""savfi-mi355x: MI355X-native MAML inner-loop adaptation path for scene-adaptive video frame
interpolation (SepConv / VoxelFlow / CAIN backbones, LSLR / Meta-SGD inner rules).

The modules mirror the reference's layout for this path (config, meta_learning_system,
inner_loop_optimizers, model_utils, loss, utils, experiment_builder, sepconv/, voxelflow/, cain/)
and call hand-written gfx950 kernels through the C ABI in include/savfi_hip.h.
"""
__version__ = "0.1.0"
