"""Dense bottleneck (reference: pasco/models/layers.py:646-726 `SPCDense3Dv2`, wired at
pasco/models/unet3d_sparse_v2.py:182-214).

Eleven dense 3-D convolutions (256 -> 256, kernels (3,3,1) (5,5,3) (7,7,5) and 1x1x1) on the
stride-8 grid (32 x 32 x 4 for a 256 x 256 x 32 scene), each followed by BatchNorm3d + ReLU.  This is
genuinely dense GEMM work.  `forward` is the plain torch formulation (reference semantics, used by
the golden tests); `forward_rows` runs the same eleven convolutions as implicit GEMMs on the
hand-written MFMA convolution kernel: the grid is a sparse tensor whose every site is active, the
(7,7,5) / (5,5,3) / (3,3,1) kernels are neighbour tables of 245 / 75 / 9 offsets (cached per grid
shape), BN + ReLU ride in the epilogue, features stay channels-last [sites, 256].
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..me.backend import ACT_RELU, MAX_KVOL, backend_for
from .fused import fold_bn


def _conv(c, k):
    pad = tuple(v // 2 for v in k)
    return nn.Sequential(nn.Conv3d(c, c, k, 1, padding=pad, bias=False), nn.Identity())


class SPCDense3Dv2(nn.Module):
    def __init__(self, init_size=16):
        super().__init__()
        c = init_size
        specs = {"a_conv1": (3, 3, 1), "a_conv2": (3, 3, 1), "a_conv3": (5, 5, 3), "a_conv4": (7, 7, 5),
                 "a_conv5": (3, 3, 1), "a_conv6": (5, 5, 3), "a_conv7": (7, 7, 5)}
        for i, (name, k) in enumerate(specs.items(), start=1):
            setattr(self, name, _conv(c, k))
            setattr(self, f"bn_{i}", nn.BatchNorm3d(c))
        self.ch_conv1 = nn.Sequential(nn.Conv3d(c, c, kernel_size=1, stride=1, bias=False), nn.Identity())
        self.bn_ch_conv1 = nn.BatchNorm3d(c)
        for i, k in enumerate(((3, 3, 1), (5, 5, 3), (7, 7, 5)), start=1):
            setattr(self, f"res_{i}", _conv(c, k))
            setattr(self, f"bn_res_{i}", nn.BatchNorm3d(c))

    def _cbr(self, conv, bn, x):
        return F.relu(bn(conv(x)))

    def forward(self, x):
        x1 = self._cbr(self.a_conv1, self.bn_1, x)
        x2 = self._cbr(self.a_conv2, self.bn_2, x1)
        x3 = self._cbr(self.a_conv3, self.bn_3, x1)
        x4 = self._cbr(self.a_conv4, self.bn_4, x1)
        t1 = x2 + x3 + x4
        x5 = self._cbr(self.a_conv5, self.bn_5, t1)
        x6 = self._cbr(self.a_conv6, self.bn_6, t1)
        x7 = self._cbr(self.a_conv7, self.bn_7, t1)
        s = x1 + x2 + x3 + x4 + x5 + x6 + x7
        y0 = self._cbr(self.ch_conv1, self.bn_ch_conv1, s)
        y1 = self._cbr(self.res_1, self.bn_res_1, x)
        y2 = self._cbr(self.res_2, self.bn_res_2, x)
        y3 = self._cbr(self.res_3, self.bn_res_3, x)
        return x1 + y0 + y1 + y2 + y3


    # ---- implicit-GEMM path on the sparse-conv kernel ---------------------------------------------------
    _KERNELS = {"a_conv1": (3, 3, 1), "a_conv2": (3, 3, 1), "a_conv3": (5, 5, 3), "a_conv4": (7, 7, 5),
                "a_conv5": (3, 3, 1), "a_conv6": (5, 5, 3), "a_conv7": (7, 7, 5), "ch_conv1": (1, 1, 1),
                "res_1": (3, 3, 1), "res_2": (5, 5, 3), "res_3": (7, 7, 5)}
    _BNS = {"a_conv1": "bn_1", "a_conv2": "bn_2", "a_conv3": "bn_3", "a_conv4": "bn_4", "a_conv5": "bn_5",
            "a_conv6": "bn_6", "a_conv7": "bn_7", "ch_conv1": "bn_ch_conv1", "res_1": "bn_res_1",
            "res_2": "bn_res_2", "res_3": "bn_res_3"}

    def _row_weight(self, name):
        """Conv3d weight [Cout, Cin, kx, ky, kz] -> [K, Cin, Cout] with K enumerated y fastest, then x, then z - the order of the
        grid's kernel maps (`CBackend.grid_offsets`: the direction in which consecutive sites are consecutive rows)."""
        w = getattr(self, name)[0].weight
        hit = getattr(self, "_roww_" + name, None)
        ver = (w._version, w.device)
        if hit is None or hit[0] != ver:
            k = w.shape[2] * w.shape[3] * w.shape[4]
            rw = w.detach().permute(4, 2, 3, 1, 0).reshape(k, w.shape[1], w.shape[0]).contiguous()
            if k == 1:
                rw = rw.reshape(w.shape[1], w.shape[0])
            hit = (ver, rw)
            object.__setattr__(self, "_roww_" + name, hit)
        return hit[1]

    def _grid_tables(self, dims, device):
        """Site coordinates + neighbour tables per kernel shape, cached per grid shape.  The sites are enumerated Z-MAJOR -
        (b, z, x, y) - not lexicographically: a run of 128 rows (one tile of the convolution kernel) then lies in one z plane
        of the 4-deep grid, and for such a tile every kernel offset that points above / below the grid (2 of the 5 dz of a
        (7, 7, 5) kernel at z = 0 or 3) has no neighbour at all - the kernel drops those stages (k_conv_dma, kvol > 27).
        -> (coords [n, 4] z-major, tables, perm [n] int64: row of the lexicographic order each z-major row is, inv: the
        inverse, the lexicographic coordinates)."""
        cache = self.__dict__.setdefault("_grid_cache", {})
        key = (tuple(int(d) for d in dims), str(device))
        if key in cache:
            return cache[key]
        b, x, y, z = key[0]
        be = backend_for(device)
        lex = torch.arange(b * x * y * z, dtype=torch.int64, device=device).view(b, x, y, z)
        perm = lex.permute(0, 3, 1, 2).reshape(-1).contiguous()            # z-major position -> lexicographic row
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), dtype=torch.int64, device=device)
        ax = [torch.arange(n, dtype=torch.int32, device=device) for n in (b, x, y, z)]
        lex_coords = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).reshape(-1, 4).contiguous()
        coords = lex_coords.index_select(0, perm).contiguous()
        tk, tv, _, _, _ = be.map_insert(coords, dedup=False)
        tables = {}
        for ks in {(3, 3, 1), (5, 5, 3), (7, 7, 5)}:
            offs = be.grid_offsets(ks)
            parts = [be.nbr_build(coords, tk, tv, offs[i:i + MAX_KVOL]) for i in range(0, len(offs), MAX_KVOL)]
            tables[ks] = torch.cat(parts, dim=0).contiguous()
        cache[key] = (coords, tables, perm, inv, lex_coords)
        return cache[key]

    def forward_rows(self, rows: torch.Tensor, dims) -> torch.Tensor:
        """rows [B*X*Y*Z, C] channels-last features of the dense grid (zeros at empty sites), sites in
        lexicographic (b,x,y,z) order -> same layout after the block.  dims = (B, X, Y, Z)."""
        assert not self.training, "inference only"
        be = backend_for(rows.device)
        _, tables, perm, inv, _ = self._grid_tables(dims, rows.device)
        n = rows.shape[0]

        from . import fused

        in_splits = {}     # operand split of each intermediate, shared by the three branches that read it

        def cbr(name, x, add=None):
            """conv - BatchNorm - ReLU of one branch; `add`: a tensor added to the result by the launch's epilogue (after the ReLU)
            - the sums of the block's branches (layers.py:700-726) ride on the convolutions instead of ten element-wise passes.
            fp32 addition is commutative: each sum below is the reference's, term by term, in its order of association."""
            ks = self._KERNELS[name]
            scale, shift = fold_bn(getattr(self, self._BNS[name]))
            w = self._row_weight(name)
            split = in_split = None
            if fused.conv_precision() == "f16x3" and be.split_supported(w.shape[-2], w.shape[-1]):
                hit = getattr(self, "_split_" + name, None)
                if hit is None or hit[0] is not w or hit[1] != fused._PRESPLIT:
                    hit = (w, fused._PRESPLIT, fused._split_of(w, be))
                    object.__setattr__(self, "_split_" + name, hit)
                split = hit[2]
                if fused._PRESPLIT:
                    if id(x) not in in_splits:
                        in_splits[id(x)] = (x, be.split_rows(x))
                    in_split = in_splits[id(x)][1]
            return be.conv_fwd(x, w, tables.get(ks), n, epi_scale=scale, epi_shift=shift, epi_act=ACT_RELU,
                               split=split, in_split=in_split, residual=add,
                               grid=(tuple(int(v) for v in dims), ks) if split is not None and ks != (1, 1, 1) else None)

        x = rows.index_select(0, perm)                     # z-major inside the block (see _grid_tables), lexicographic outside
        x1 = cbr("a_conv1", x)
        x2 = cbr("a_conv2", x1)
        t1 = cbr("a_conv4", x1, add=cbr("a_conv3", x1, add=x2))                      # (x2 + x3) + x4
        s = cbr("a_conv7", t1, add=cbr("a_conv6", t1, add=cbr("a_conv5", t1, add=x1 + t1)))      # (((x1 + t1) + x5) + x6) + x7
        y = cbr("ch_conv1", s, add=x1)                                                # x1 + y0
        y = cbr("res_3", x, add=cbr("res_2", x, add=cbr("res_1", x, add=y)))          # ((. + y1) + y2) + y3
        return y.index_select(0, inv)
