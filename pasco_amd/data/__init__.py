"""Data formats either side of the hot path: SemanticKITTI bit-packed voxel files, point
labels, WaffleIron feature pickles, instance-label pickles -> the input contract of `PascoNet.step_inference`
(a0), and Lightning checkpoints -> `PascoNet`."""
from .semantic_kitti import (build_item, collate, pack_bits, read_instance_label_pickle, read_invalid, read_label,
                             read_occluded, read_occupancy, read_point_instance_labels, read_pointcloud,
                             read_waffleiron_features, transform_coords, transform_scene, unpack_bits, FrameReader)
from .checkpoint import load_lightning_state_dict, net_from_checkpoint, remap_reference_state_dict

__all__ = ["build_item", "collate", "pack_bits", "unpack_bits", "read_instance_label_pickle", "read_invalid",
           "read_label", "read_occluded", "read_occupancy", "read_point_instance_labels", "read_pointcloud",
           "read_waffleiron_features", "transform_coords", "transform_scene", "FrameReader",
           "load_lightning_state_dict", "net_from_checkpoint", "remap_reference_state_dict"]
