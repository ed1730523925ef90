"""pasco_amd - MI355X-native sparse-voxel panoptic scene completion engine (PaSCo hot path).

`pasco_amd.me`     MinkowskiEngine operator surface on top of the C ABI (include/pasco_hip.h)
`pasco_amd.graph`  PaSCo's sparse U-Net + mask-transformer graph restated on fused HIP launches
"""
__version__ = "0.1.0"
