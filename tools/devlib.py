"""The DEVELOPMENT build of the library for the experiment tools: `pasco_amd.build.build_hip(dev=True)` compiles the sources with
-DPH_DEV into pasco_amd/csrc/libpascohip_dev.so - the product kernels plus what the product library does not carry (round 6):
PASCO_* environment switches inside the dispatch, `ph_conv_dma_set_ablate`, `ph_conv_dma_occupancy`, `ph_dma_trace_read`,
`ph_wop_trace_enable / _read`, `ph_conv_lin_set`.  Call `use_dev_library()` BEFORE the first `hip_backend()`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def use_dev_library():
    from pasco_amd.build import build_hip
    from pasco_amd.me import backend
    assert backend._hip_backend is None, "use_dev_library() must run before the first hip_backend()"
    backend.HIP_LIB_PATH = build_hip(dev=True, verbose=False)
    return backend.HIP_LIB_PATH
