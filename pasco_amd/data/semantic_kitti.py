"""SemanticKITTI / PaSCo on-disk formats -> the a0 input contract of the inference path.

Restates, file format by file format, what the reference's data layer reads (no training-side label pyramids,
no augmentation sampling - the caller passes the rigid transform T of each subnet):

  bit-packed voxel grids   pasco/data/semantic_kitti/io_data.py:11-24 (unpack), :35-45 (pack),
                           :114-138 (.bin occupancy, .label uint16, .invalid / .occluded)
  velodyne points          io_data.py:146-150 (float32 x, y, z, remission)
  per-point labels         pasco/data/semantic_kitti/kitti_dataset.py:318-328 (int32, low 16 bits = class)
  WaffleIron features      kitti_dataset.py:290-303 (pickle: embedding [E, 256, P], coords [P, 4], vote [P, V])
  instance labels          kitti_dataset.py:329-339 (pickle: semantic_labels / instance_labels uint8 grids)
  frame -> subnet item     kitti_dataset.py:339-455 (extent crop, radius, voxelise, transform) and :150-175
                           (min_C floored to the completion scale, max_C)
  items -> batch           pasco/data/semantic_kitti/collate.py:76-105 (global bounds rounded up to the scale)
  rigid transforms         pasco/models/transform_utils.py:60-74 (`transform`), :123-161 (`transform_scene`)

Pickles are Python pickles: load them only from sources you trust (the reference's own preprocessing output).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

VOX_ORIGIN = np.array([0.0, -25.6, -2.0])
VOXEL_SIZE = 0.2
MAX_EXTENT = (51.2, 25.6, 4.4)          # kitti_dataset.py:58 (sic: 4.4, not 4.0)
MIN_EXTENT = (0.0, -25.6, -2.0)


# ---- bit-packed voxel files ---------------------------------------------------------------------------
def unpack_bits(compressed: np.ndarray) -> np.ndarray:
    """uint8 [n] -> uint8 [8 n] of 0 / 1, most significant bit first (io_data.py:11-24)."""
    return np.unpackbits(np.ascontiguousarray(compressed, dtype=np.uint8))


def pack_bits(array: np.ndarray) -> np.ndarray:
    """0 / 1 array -> bitwise uint8, most significant bit first (io_data.py:35-45); the size must divide by 8."""
    flat = np.asarray(array).reshape(-1)
    if flat.size % 8 != 0:
        raise ValueError("pack_bits: number of voxels must be a multiple of 8")
    return np.packbits(flat.astype(bool).astype(np.uint8))


def _read(path: str, dtype, do_unpack: bool) -> np.ndarray:
    raw = np.fromfile(path, dtype=dtype)
    return unpack_bits(raw) if do_unpack else raw


def read_occupancy(path: str) -> np.ndarray:
    """voxels/*.bin: bit-packed occupancy -> float32 [X*Y*Z] (io_data.py:136-138)."""
    return _read(path, np.uint8, True).astype(np.float32)


def read_label(path: str) -> np.ndarray:
    """voxels/*.label: uint16 per voxel -> float32 (io_data.py:121-123)."""
    return _read(path, np.uint16, False).astype(np.float32)


def read_invalid(path: str) -> np.ndarray:
    """voxels/*.invalid: bit-packed mask -> uint8 (io_data.py:126-128)."""
    return _read(path, np.uint8, True)


def read_occluded(path: str) -> np.ndarray:
    """voxels/*.occluded: bit-packed mask -> uint8 (io_data.py:131-133)."""
    return _read(path, np.uint8, True)


def read_pointcloud(path: str) -> np.ndarray:
    """velodyne/*.bin -> float32 [N, 4] = x, y, z, remission (io_data.py:146-150)."""
    return _read(path, np.float32, False).reshape(-1, 4)


def read_point_instance_labels(path: str) -> np.ndarray:
    """labels/*.label: int32 per point, low 16 bits kept -> int32 [N, 1] (kitti_dataset.py:318-328)."""
    return (np.fromfile(path, dtype=np.int32).reshape(-1, 1) & 0xFFFF)


# ---- pickles ---------------------------------------------------------------------------------------------
def read_waffleiron_features(path: str, embedding_index: Optional[int] = None,
                             rng: Optional[np.random.Generator] = None):
    """seg_feats_tta/*.pkl -> (xyz [P,3], vote [P,V], intensity [P,1], embedding [P,256]) (kitti_dataset.py:290-303).
    The file holds E test-time-augmented embeddings; the reference draws one at random per call - pass
    `embedding_index` for a reproducible choice, or an `rng`."""
    with open(path, "rb") as f:
        data = pickle.load(f)
    emb = data["embedding"]
    if embedding_index is None:
        embedding_index = int((rng or np.random.default_rng()).integers(0, emb.shape[0]))
    embedding = emb[embedding_index].T
    xyz_density = data["coords"]
    return xyz_density[:, :3], data["vote"], xyz_density[:, 3:], embedding


def read_instance_label_pickle(path: str):
    """instance_labels_v2/<seq>/<frame>_1_<down>.pkl -> (semantic uint8 grid, instance uint8 grid)
    (kitti_dataset.py:329-339)."""
    with open(path, "rb") as f:
        data = pickle.load(f)
    return data["semantic_labels"].astype(np.uint8), data["instance_labels"].astype(np.uint8)


# ---- rigid transforms on the voxel lattice --------------------------------------------------------------
def transform_coords(coords: torch.Tensor, T: torch.Tensor, resolution: float = 0.2) -> torch.Tensor:
    """Voxel indices -> metres (voxel centres) -> T -> voxel indices, rounded to nearest even like torch.round
    (transform_utils.py:60-74).  Returns int32."""
    min_bound = torch.tensor([0, -25.6, -2]).reshape(1, 3).to(coords.device)
    pts = coords * resolution + resolution / 2
    pts = min_bound + pts
    hom = torch.cat([pts, torch.ones(pts.shape[0], 1, device=pts.device)], dim=1).type_as(T)
    new = (T @ hom.T).T[:, :3]
    new = (new - min_bound - resolution / 2) / resolution
    return torch.round(new).int()


def transform_scene(from_coords: torch.Tensor, T: torch.Tensor, voxel_features: torch.Tensor, to_coords_bnd=None):
    """Resample a labelled grid under T without holes (transform_utils.py:123-161): the bounding box of the
    transformed coordinates is sampled densely, every sample is projected back with T^-1 and reads the source
    voxel it lands in (zero outside).  The reference does the read with `grid_sample(mode="nearest",
    align_corners=True)` on integer coordinates, which is plain indexing - done here as such.
    voxel_features [F, H, W, D] -> (features [N, F], coords int32 [N, 3], (min, max))."""
    if to_coords_bnd is None:
        to = transform_coords(from_coords, T)
        to_coords_bnd = (to.min(0)[0], to.max(0)[0])
    mn, mx = to_coords_bnd
    size = (mx - mn + 1).tolist()
    # the reference enumerates np.meshgrid(x, y, z) with the default 'xy' indexing: y outermost, then x, then z
    gx, gy, gz = np.meshgrid(np.arange(size[0]), np.arange(size[1]), np.arange(size[2]))
    grid = torch.from_numpy(np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(float)).type_as(from_coords)
    to_coords = grid + mn.reshape(1, 3)
    back = transform_coords(to_coords, torch.inverse(T)).long()
    Fch, H, W, D = voxel_features.shape
    ok = (back[:, 0] >= 0) & (back[:, 0] < H) & (back[:, 1] >= 0) & (back[:, 1] < W) & (back[:, 2] >= 0) & (back[:, 2] < D)
    out = torch.zeros((back.shape[0], Fch), dtype=torch.float32)
    b = back[ok]
    out[ok] = voxel_features[:, b[:, 0], b[:, 1], b[:, 2]].T.float()
    return out, to_coords.int(), to_coords_bnd


def compute_scene_size(min_c: torch.Tensor, max_c: torch.Tensor, scale: int = 1) -> torch.Tensor:
    """pasco/models/misc.py:30-32."""
    return (torch.ceil((max_c - min_c + 1) / scale) * scale).int()


# ---- one frame -> one subnet item ---------------------------------------------------------------------------
def build_item(xyz: np.ndarray, vote: np.ndarray, intensity: np.ndarray, embedding: np.ndarray,
               semantic_label: np.ndarray, instance_label: np.ndarray, T: Optional[torch.Tensor] = None,
               complete_scale: int = 8, point_labels: Optional[np.ndarray] = None) -> Dict:
    """The inference-side fields of `KittiDataset.get_individual` (kitti_dataset.py:339-455,150-175) for one
    subnet: point features [P, V + 1 + 1 + 256 + 6] (votes, intensity, radius, embedding, offset to the voxel
    centre, xyz), integer voxel coordinates under T, the transform, and the completion bounds min_C / max_C taken
    from the transformed label grids."""
    T = torch.eye(4) if T is None else T
    keep = ((xyz[:, 0] < MAX_EXTENT[0]) & (xyz[:, 0] >= MIN_EXTENT[0]) & (xyz[:, 1] < MAX_EXTENT[1])
            & (xyz[:, 1] >= MIN_EXTENT[1]) & (xyz[:, 2] < MAX_EXTENT[2]) & (xyz[:, 2] >= MIN_EXTENT[2]))
    xyz = xyz[keep]
    vote_intensity = np.concatenate((vote, intensity), axis=1)[keep]
    embedding = embedding[keep]
    if point_labels is not None:
        point_labels = point_labels[keep]

    sem = torch.from_numpy(semantic_label)
    sem_coords = torch.nonzero(sem != 255)
    sem_sparse, sem_coords, bnd = transform_scene(sem_coords, T, sem.unsqueeze(0) + 1)
    nz = sem_sparse.sum(dim=1) != 0
    sem_sparse, sem_coords = sem_sparse[nz] - 1, sem_coords[nz]
    ins = torch.from_numpy(instance_label)
    ins_coords = torch.nonzero(ins)
    if ins_coords.shape[0] > 0:
        ins_sparse, ins_coords, _ = transform_scene(ins_coords, T, ins.unsqueeze(0) + 1, to_coords_bnd=bnd)
    else:
        ins_sparse, ins_coords = torch.zeros((0, 1)), torch.zeros((0, 3)).long()
    nz = ins_sparse.sum(dim=1) != 0
    ins_sparse, ins_coords = ins_sparse[nz] - 1, ins_coords[nz]

    radius = np.linalg.norm(xyz, axis=1)[..., np.newaxis]
    feat = np.concatenate((vote_intensity, radius, embedding), axis=1)
    origin = VOX_ORIGIN.reshape(1, 3)
    coords = (xyz - origin) // VOXEL_SIZE
    centres = (coords.astype(np.float32) + 0.5) * VOXEL_SIZE + origin
    return_xyz = np.concatenate((xyz - centres, xyz), axis=1)
    in_feat = torch.from_numpy(np.concatenate([feat, return_xyz], axis=1)).float()
    in_coord = transform_coords(torch.from_numpy(coords), T).long()

    min_c, max_c = sem_coords.min(dim=0)[0], sem_coords.max(dim=0)[0]
    if ins_coords.shape[0] > 0:
        min_c = torch.min(min_c, ins_coords.min(dim=0)[0])
        max_c = torch.max(max_c, ins_coords.max(dim=0)[0])
    min_c = (torch.floor(min_c.float() / complete_scale) * complete_scale).int()
    max_c = torch.ceil(max_c)
    return {"in_feat": in_feat, "in_coord": in_coord, "T": T, "min_C": min_c, "max_C": max_c,
            "xyz": xyz - origin, "semantic_label_sparse": (sem_sparse.to(torch.uint8), sem_coords),
            "instance_label_sparse": (ins_sparse.to(torch.uint8), ins_coords),
            "input_pcd_instance_label": None if point_labels is None else torch.from_numpy(point_labels)}


def collate(items: Sequence[Dict], complete_scale: int = 8) -> Dict:
    """Subnet items -> the batch `Net.step_inference` consumes (collate.py:76-105): per-subnet lists plus the global
    bounds, the extent rounded up to a multiple of the completion scale."""
    min_cs = [it["min_C"] for it in items]
    max_cs = [it["max_C"] for it in items]
    gmin = torch.min(torch.stack(min_cs), dim=0)[0]
    gmax = torch.max(torch.stack(max_cs), dim=0)[0]
    gmax = gmin + compute_scene_size(gmin, gmax, scale=complete_scale) - 1     # inclusive
    return {"in_feats": [it["in_feat"] for it in items], "in_coords": [it["in_coord"] for it in items],
            "Ts": [it["T"] for it in items], "min_Cs": min_cs, "max_Cs": max_cs,
            "global_min_Cs": gmin, "global_max_Cs": gmax, "xyz": [it["xyz"] for it in items],
            "input_pcd_instance_label": [it.get("input_pcd_instance_label") for it in items]}


class FrameReader:
    """Directory layout of the reference's SemanticKITTI setup (kitti_dataset.py:100-112,344-370):
        <root>/dataset/sequences/<seq>/labels/<frame>.label
        <preprocess_root>/instance_labels_v2/<seq>/<frame>_1_1.pkl
        <preprocess_root>/waffleiron_v2/sequences/<seq>/seg_feats_tta/<frame>.pkl
    `batch(seq, frame, Ts)` returns the collated a0 contract for len(Ts) subnets (the reference's validation loader
    feeds every subnet the same frame under its own transform)."""

    def __init__(self, root: str, preprocess_root: str, complete_scale: int = 8):
        self.root, self.preprocess_root, self.complete_scale = root, preprocess_root, complete_scale

    def paths(self, sequence: str, frame_id: str):
        return (os.path.join(self.preprocess_root, "instance_labels_v2", sequence, f"{frame_id}_1_1.pkl"),
                os.path.join(self.preprocess_root, "waffleiron_v2/sequences", sequence, "seg_feats_tta", f"{frame_id}.pkl"),
                os.path.join(self.root, "dataset", "sequences", sequence, "labels", f"{frame_id}.label"))

    def batch(self, sequence: str, frame_id: str, Ts: Sequence[torch.Tensor], embedding_index: int = 0) -> Dict:
        lab, feats, pts = self.paths(sequence, frame_id)
        sem, ins = read_instance_label_pickle(lab)
        xyz, vote, intensity, emb = read_waffleiron_features(feats, embedding_index=embedding_index)
        plab = read_point_instance_labels(pts) if os.path.exists(pts) else None
        items = [build_item(xyz, vote, intensity, emb, sem, ins, T, self.complete_scale, plab) for T in Ts]
        return collate(items, self.complete_scale)
