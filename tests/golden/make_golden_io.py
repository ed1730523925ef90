"""Fixtures for the data-format layer, produced by the REFERENCE's own readers.

Runs only in the build container (needs /root/reference).  Step 1 writes tiny synthetic input files in the
reference's on-disk formats under tests/golden/kitti_mini/ (data, generated with numpy - not reference code);
step 2 imports the reference (`pasco.data.semantic_kitti.io_data`, `KittiDataset.get_individual`, `collate_fn`,
`transform_scene`) and runs it on those files; step 3 stores what the reference returned in tests/golden/io_*.npz;
step 4 writes a Lightning-shaped checkpoint whose state dict comes from the reference's own modules
(`TransformerPredictorV2`, `UNet3DV2`, `CylinderFeat`, a `criterion.*` buffer) with the key prefixes `Net` gives them.

    python tests/golden/make_golden_io.py
"""
import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up sys.path, the MinkowskiEngine alias and the inert stubs)

MINI = os.path.join(HERE, "kitti_mini")
GRID = (64, 64, 16)
SEQ, FRAME = "08", "000005"


def write_inputs():
    rng = np.random.default_rng(7)
    vox = os.path.join(MINI, "dataset", "sequences", SEQ, "voxels")
    lab = os.path.join(MINI, "dataset", "sequences", SEQ, "labels")
    ins = os.path.join(MINI, "preprocess", "instance_labels_v2", SEQ)
    wfl = os.path.join(MINI, "preprocess", "waffleiron_v2", "sequences", SEQ, "seg_feats_tta")
    for d in (vox, lab, ins, wfl, os.path.join(MINI, "dataset", "sequences", SEQ, "velodyne")):
        os.makedirs(d, exist_ok=True)
    n = GRID[0] * GRID[1] * GRID[2]
    occ = rng.random(n) < 0.1
    np.packbits(occ.astype(np.uint8)).tofile(os.path.join(vox, FRAME + ".bin"))
    np.packbits((rng.random(n) < 0.3).astype(np.uint8)).tofile(os.path.join(vox, FRAME + ".invalid"))
    np.packbits((rng.random(n) < 0.2).astype(np.uint8)).tofile(os.path.join(vox, FRAME + ".occluded"))
    (rng.integers(0, 260, n).astype(np.uint16) * occ).astype(np.uint16).tofile(os.path.join(vox, FRAME + ".label"))
    # labelled completion grids: a ground sheet, two boxes (one of them a "thing" with an instance id), 255 = unknown
    sem = np.full(GRID, 255, np.uint8)
    sem[4:60, 6:58, :] = 0
    sem[4:60, 6:58, 3:5] = 9
    sem[20:28, 20:30, 5:10] = 1
    sem[40:44, 30:33, 5:8] = 6
    inst = np.zeros(GRID, np.uint8)
    inst[20:28, 20:30, 5:10] = 1
    inst[40:44, 30:33, 5:8] = 2
    with open(os.path.join(ins, f"{FRAME}_1_1.pkl"), "wb") as f:
        pickle.dump({"semantic_labels": sem, "instance_labels": inst}, f)
    P, V, E = 500, 19, 3
    xyz = np.stack([rng.uniform(-2, 14, P), rng.uniform(-7, 7, P), rng.uniform(-2.5, 1.5, P)], 1).astype(np.float32)
    with open(os.path.join(wfl, FRAME + ".pkl"), "wb") as f:
        pickle.dump({"embedding": rng.standard_normal((E, 256, P)).astype(np.float32),
                     "coords": np.concatenate([xyz, rng.random((P, 1)).astype(np.float32)], 1),
                     "vote": rng.random((P, V)).astype(np.float32)}, f)
    (rng.integers(0, 1 << 20, P).astype(np.int32)).tofile(os.path.join(lab, FRAME + ".label"))
    np.concatenate([xyz, rng.random((P, 1)).astype(np.float32)], 1).astype(np.float32).tofile(
        os.path.join(MINI, "dataset", "sequences", SEQ, "velodyne", FRAME + ".bin"))


def golden_io():
    import pasco.data.semantic_kitti.io_data as IO
    vox = os.path.join(MINI, "dataset", "sequences", SEQ, "voxels")
    occ = IO._read_occupancy_SemKITTI(os.path.join(vox, FRAME + ".bin"))
    rng = np.random.default_rng(11)
    bits = (rng.random(4096) < 0.5).astype(np.uint8)
    G.save("io_files.npz",
           occupancy=occ, label=IO._read_label_SemKITTI(os.path.join(vox, FRAME + ".label")),
           invalid=IO._read_invalid_SemKITTI(os.path.join(vox, FRAME + ".invalid")),
           occluded=IO._read_occluded_SemKITTI(os.path.join(vox, FRAME + ".occluded")),
           pointcloud=IO._read_pointcloud_SemKITTI(os.path.join(MINI, "dataset", "sequences", SEQ, "velodyne", FRAME + ".bin")),
           bits=bits, packed=IO.pack(bits), unpacked=IO.unpack(IO.pack(bits)))


def golden_items():
    """`KittiDataset.get_individual` + `collate_fn` on the mini frame: identity transform and a fixed rigid one."""
    import pasco.data.semantic_kitti.kitti_dataset as KD
    from pasco.data.semantic_kitti.collate import collate_fn
    from pasco.models.transform_utils import generate_transformation
    from pasco.data.semantic_kitti.params import thing_ids
    T_fixed = generate_transformation(rot=17.0, translation=(0.4, -0.3, 0.1), flip_dim=1, scale=1.0)
    out = {}
    items = []
    for tag, T in (("eye", None), ("rigid", T_fixed)):
        ds = object.__new__(KD.KittiDataset)
        ds.root = MINI
        ds.preprocess_root = os.path.join(MINI, "preprocess")
        ds.instance_label_root = os.path.join(ds.preprocess_root, "instance_labels_v2")
        ds.complete_scale = 8
        ds.data_aug = T is not None
        ds.max_angle, ds.scale_range, ds.max_translation = 0.0, 0.0, np.zeros(3)
        ds.split = "val"
        ds.n_subnets = 1
        ds.n_fuse_scans = 1
        ds.max_extent = (51.2, 25.6, 4.4)
        ds.min_extent = np.array([0, -25.6, -2.0])
        ds.vox_origin = np.array([0, -25.6, -2])
        ds.voxel_size = 0.2
        ds.thing_ids = thing_ids
        ds.scans = [{"sequence": SEQ, "frame_id": FRAME}]
        ds.poses = {int(SEQ): [np.eye(4)] * 16}
        if T is not None:
            KD.generate_random_transformation = lambda **kw: T_fixed
        np.random.seed(0)      # load_file draws the embedding index with np.random.randint
        state = np.random.get_state()
        emb_index = int(np.random.randint(0, 3))
        np.random.set_state(state)
        item = ds.get_individual(0)
        items.append(item)
        out.update({f"{tag}_in_feat": item["in_feat"], f"{tag}_in_coord": item["in_coord"], f"{tag}_T": item["T"],
                    f"{tag}_min_C": item["min_C"], f"{tag}_max_C": item["max_C"], f"{tag}_xyz": item["xyz"],
                    f"{tag}_emb_index": np.array(emb_index)})
    batch = collate_fn(items, 8)
    out.update(global_min_Cs=batch["global_min_Cs"], global_max_Cs=batch["global_max_Cs"], T_fixed=T_fixed)
    # transform_scene on its own
    from pasco.models.transform_utils import transform_scene
    g = torch.Generator().manual_seed(3)
    grid = (torch.rand((2, 12, 10, 6), generator=g) > 0.6).float() * torch.randint(1, 9, (2, 12, 10, 6), generator=g)
    coords = torch.nonzero(grid[0] != 0)
    f, c, bnd = transform_scene(coords, T_fixed, grid)
    out.update(ts_grid=grid, ts_coords=coords, ts_feat=f, ts_out_coords=c, ts_bnd_min=bnd[0], ts_bnd_max=bnd[1])
    G.save("io_items.npz", **out)


def golden_checkpoint():
    """A checkpoint in the layout Lightning writes for the reference `Net` (net_panoptic_sparse.py:108-175)."""
    from pasco.models.unet3d_sparse_v2 import UNet3DV2, CylinderFeat
    from pasco.models.transformer.transformer_predictor_v2 import TransformerPredictorV2
    torch.manual_seed(21)
    f, n_infers, nq, in_ch = 8, 2, 6, 283      # the frame files carry 283-channel point features
    tp = TransformerPredictorV2(dropout=0.0, nheads=8, hidden_dim=48, enc_layers=0, num_queries=nq, dim_feedforward=96,
                                dec_layers=1, aux_loss=False, mask_dim=f, n_infers=n_infers, query_sample_ratio=1.0,
                                in_channels=[f * 4, f * 2, f])
    unet = UNet3DV2(heavy_decoder=False, drop_path_rate=0.0, n_classes=20, in_channels=f * n_infers,
                    transformer_predictor=tp, f_maps=[f, f * 2, f * 4, f * 4], dense3d_dropout=0.0, n_infers=n_infers,
                    decoder_dropouts=[0.0] * 3, num_queries=nq, query_sample_ratio=1.0, encoder_dropouts=[0.0] * 3,
                    use_se_layer=False)
    feat = CylinderFeat(fea_dim=in_ch, out_pt_fea_dim=f)
    gen = torch.Generator().manual_seed(5)
    for m in (unet, feat):
        G.randomise_bn(m, gen)

    class NetShell(torch.nn.Module):      # the attribute names `Net.__init__` registers its stateful children under
        def __init__(self):
            super().__init__()
            self.transformer_predictor = tp
            self.unet3d = unet
            self.feat = feat
            self.criterion = torch.nn.Module()
            self.criterion.register_buffer("empty_weight", torch.ones(21))

    shell = NetShell()
    sd = shell.state_dict()       # keeps every alias of the shared predictor, as Lightning's checkpoint does
    ckpt = {"epoch": 3, "global_step": 1234, "pytorch-lightning_version": "2.0.0", "state_dict": sd,
            "hyper_parameters": {"n_classes": 20, "n_infers": n_infers, "in_channels": in_ch, "f": f, "num_queries": nq,
                                 "heavy_decoder": False, "iou_threshold": 0.2, "overlap_threshold": 0.4,
                                 "object_mask_threshold": 0.7, "class_frequencies": np.ones(20), "lr": 1e-4}}
    torch.save(ckpt, os.path.join(HERE, "net_mini.ckpt"))
    print("checkpoint keys:", len(sd), "aliases:", sum(k.startswith("unet3d.decoder_generative.transformer_predictor.") for k in sd))


if __name__ == "__main__":
    write_inputs()
    golden_io()
    golden_items()
    golden_checkpoint()
