"""pointwise_amd -- MI355X-native conv3p (pointwise convolution) hot path of hkust-vgd/pointwise.

Only what the path needs: csrc/ (hand-written gfx950 HIP kernels + the C ABI of include/conv3p.h),
conv3p_op (host mirror of the reference's operator interface), stack (the models' conv3p layer stacks),
distributed (batch sharding + RCCL all-reduce of the weight gradients), synth (synthetic clouds).
"""
from .conv3p_op import (Conv3pFunction, Conv3pInvalidArgument, Conv3pRuntimeError, conv3p, conv3p_autograd,
                        conv3p_grad, conv3p_layer, conv3p_layer_grad, neighbor_count, selu, selu_grad)

__all__ = ["conv3p", "conv3p_grad", "conv3p_layer", "conv3p_layer_grad", "conv3p_autograd", "Conv3pFunction", "neighbor_count", "selu", "selu_grad",
           "Conv3pInvalidArgument", "Conv3pRuntimeError"]
