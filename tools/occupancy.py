"""Resident workgroups per CU of the convolution kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pasco_amd.me.backend import hip_backend
torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import use_dev_library   # noqa: E402
use_dev_library()      # the hooks below exist only in the development build (-DPH_DEV)
lib = hip_backend().lib
lib.ph_conv_dma_occupancy.argtypes = [C.c_int]
for i, name in enumerate(("k_conv_dma<4,2,2,2,2,false>", "k_conv_dma<4,2,2,2,2,true>", "k_conv_dma<4,4,1,1,2,false>", "k_conv_dma<4,4,1,1,1,false>")):
    print(name, lib.ph_conv_dma_occupancy(i))
