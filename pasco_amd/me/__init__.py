"""`pasco_amd.me` - the MinkowskiEngine operator surface PaSCo uses, served by libpascohip.so.

Drop-in use:  `import pasco_amd.me as ME`  or
`sys.modules["MinkowskiEngine"] = pasco_amd.me` before importing `pasco.models.*`.
"""
from enum import Enum

from . import utils  # noqa: F401
from .backend import hip_backend, backend_for  # noqa: F401
from .core import (CoordinateManager, CoordinateMapKey, SparseTensor, TensorField, to_sparse,  # noqa: F401
                   kernel_offsets)
from .modules import (MinkowskiModuleBase, MinkowskiConvolution, MinkowskiConvolutionTranspose,  # noqa: F401
                      MinkowskiGenerativeConvolutionTranspose, MinkowskiBatchNorm,
                      MinkowskiSyncBatchNorm, MinkowskiReLU, MinkowskiLeakyReLU, MinkowskiSigmoid,
                      MinkowskiSoftmax, MinkowskiELU, MinkowskiDropout, MinkowskiLinear,
                      MinkowskiPruning, MinkowskiMaxPooling, MinkowskiGlobalPooling,
                      MinkowskiGlobalAvgPooling, MinkowskiBroadcastMultiplication,
                      MinkowskiChannelwiseConvolution, MinkowskiPoolingTranspose, MinkowskiAvgPooling,
                      cat)


class SparseTensorQuantizationMode(Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class MinkowskiAlgorithm(Enum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


__version__ = "0.5.4+pasco_amd"
