"""Mask-transformer decoder over sparse voxels (reference:
pasco/models/transformer/transformer_predictor_v2.py:11-303, blocks.py:9-138,
position_encoding.py:71-135).

Three decoder layers over the voxel sets at tensor stride 4, 2, 1: masked cross-attention
(queries -> voxels), self-attention, FFN; class / mask heads before the first and after every layer.
State-dict keys follow the reference.  Quirks reproduced on purpose:
  * the sine position encoding normalises x / (x + 1e-6) * 2*pi, i.e. ~2*pi for any non-zero
    coordinate and 0 for 0 (position_encoding.py:100-104);
  * zero-padded voxel rows take part as attention keys (padding_mask=None,
    transformer_predictor_v2.py:175-177) and are looked up with wrap-around indices;
  * a query masked everywhere attends everywhere (:163-164);
  * cross-attention / FFN add their residual to the *normed* tensor (blocks.py:82,91,118-120).
The attention-mask construction avoids the reference's dense [1,100,X,Y,Z] detour: the pooled
keep-mask stays sparse and level voxels look their parent block up in its hash map.
"""
from __future__ import annotations

import math
import os
import threading
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import me as ME
from ..me.backend import backend_for
from . import fused as fused_mod
from .fused import batched_rows_matmul, linear_rows, prepare_batched_weights, split_rows_2d

def _unsplit_rows(op: torch.Tensor, c: int) -> torch.Tensor:
    """fp32 rows a split operand [n, c/32, 2, 32] stands for (diagnostic / fallback path)."""
    from ..me.backend import SPLIT_ACT_EXP2
    x = op.float()
    return (x[:, :, 0] + x[:, :, 1]).reshape(op.shape[0], -1)[:, :c] * float(2.0 ** -SPLIT_ACT_EXP2)


# torch registers generator / allocator state per capture in process-wide tables: one capture at a time
_CAPTURE_LOCK = threading.Lock()


def sine_position_encoding(coords: torch.Tensor, num_pos_feats: int, temperature: float = 10000.0,
                           scale: float = 2 * math.pi) -> torch.Tensor:
    """coords [N,3] -> [N, 3*num_pos_feats]; interleaved sin/cos per axis, axes concatenated."""
    c = coords.float()
    c = c / (c + 1e-6) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32, device=coords.device)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    ang = c[:, :, None] / dim_t                                   # [N,3,F]
    enc = torch.stack((ang[:, :, 0::2].sin(), ang[:, :, 1::2].cos()), dim=2)   # [N,3,2,F/2]
    return enc.flatten(2).flatten(1)


class PositionEmbeddingSineSparse(nn.Module):
    TABLE_LO, TABLE_HI = -1024, 4096     # coordinate values served by lookup (others are evaluated)

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        assert normalize, "the served configuration uses normalize=True"
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.scale = 2 * math.pi if scale is None else scale

    def dim_t(self, device) -> torch.Tensor:
        hit = self.__dict__.get("_dim_t")
        if hit is None or hit.device != device:
            i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            hit = (self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)).contiguous()
            fused_mod.publish(hit)
            self.__dict__["_dim_t"] = hit
        return hit

    def table(self, device) -> torch.Tensor:
        """[T, f] encodings of one axis for the integer values TABLE_LO .. TABLE_HI - 1 (the encoding of an axis depends
        on that axis' integer value only; evaluated by the same entry point as the rows, so a lookup returns exactly
        what the evaluation would)."""
        tab = self.__dict__.get("_table")
        if tab is None or tab.device != device:      # 2.6 MB, built once
            tab = backend_for(device).sine_pe_table(self.dim_t(device), self.scale, self.TABLE_LO, self.TABLE_HI)
            fused_mod.publish(tab)
            self.__dict__["_table"] = tab
        return tab

    EPS_EXP2 = 14        # the small angle offsets (<= 7e-6) are stored as eps * 2^14 so that they are normal f16 values

    def angle_model(self, device):
        """What a position term needs to know about a coordinate value (`ph_attn_cross_feat`): the encoding's angle is
        c / (c + 1e-6) * scale in fp32 (position_encoding.py:100-104) = 0 for c = 0, exactly `scale` for every |c| from a few
        dozen up (the ratio rounds to 1), and scale + eps_c in between.  -> (eps [T] fp32 = eps_c * 2^EPS_EXP2 with eps of the
        value 0 set to 0, G [f] fp64 = d table / d angle at `scale`, index of the value 0, index of a far value)."""
        hit = self.__dict__.get("_angle_model")
        if hit is not None and hit[0].device == device:
            return hit
        lo, hi = self.TABLE_LO, self.TABLE_HI
        c = torch.arange(lo, hi, dtype=torch.float32, device=device)
        v = c / (c + 1e-6) * self.scale                                   # the reference's own fp32 sequence
        far = torch.tensor(self.scale, dtype=torch.float32, device=device)
        eps = (v.double() - far.double())
        eps[-lo] = 0.0                                                     # the value 0 has its own column
        d = self.dim_t(device).double()
        half = self.num_pos_feats // 2
        ang = far.double() / torch.cat([d[0::2], d[1::2]])                 # table layout: sin block, then cos block
        G = torch.cat([ang[:half].cos() / d[0::2], -ang[half:].sin() / d[1::2]])
        i_far = hi - lo - 1
        assert float(eps[i_far]) == 0.0 and float(eps[0]) == 0.0, "the far angle must be exact at both ends of the table"
        hit = ((eps * float(2 ** self.EPS_EXP2)).float().contiguous(), G, -lo, i_far)
        fused_mod.publish(hit[0])
        self.__dict__["_angle_model"] = hit
        return hit

    def block_table(self, device) -> torch.Tensor:
        """[3, T, 3 f]: axis a's table in channel block a, zeros elsewhere - the encoding itself as a per-axis table
        residual (ph_conv_desc.axis_table)."""
        bt = self.__dict__.get("_block_table")
        if bt is None or bt.device != device:
            tab = self.table(device)
            T, f = tab.shape
            bt = torch.zeros((3, T, 3 * f), dtype=tab.dtype, device=device)
            for a in range(3):
                bt[a, :, a * f:(a + 1) * f] = tab
            fused_mod.publish(bt)
            self.__dict__["_block_table"] = bt.contiguous()
            bt = self.__dict__["_block_table"]
        return bt

    def forward(self, coords, coff: int = 0):
        """coords [N, >=3] integer rows with x,y,z from column `coff`.  On a device with a backend: one kernel
        (ph_sine_pe); otherwise the torch formula."""
        if coords.dtype == torch.int32 and coords.is_contiguous() and (coords.is_cuda or _has_checker()):
            be = backend_for(coords.device)
            dim_t = self.dim_t(coords.device)
            return be.sine_pe(coords, dim_t, self.scale, coff, table=self.table(coords.device), tab_lo=self.TABLE_LO)
        return sine_position_encoding(coords[:, coff:coff + 3], self.num_pos_feats, self.temperature, self.scale)


def _has_checker() -> bool:
    from ..me import backend
    return backend._checker_backend is not None


def _xavier(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


def _attention_math(q, k, v, bias=None):
    """softmax(q k^T / sqrt(dh) + bias) v for [B, H, Q, dh] x [B, H, N, dh] in plain fp32 matmuls (what
    `F.scaled_dot_product_attention`'s math backend computes; the fused backends are never selected here)."""
    att = torch.matmul(q * (float(q.shape[-1]) ** -0.5), k.transpose(-1, -2))
    if bias is not None:
        att = att + bias
    return torch.matmul(torch.softmax(att, dim=-1), v)


_LEAD_PATTERNS = {}


def _lead_pattern(lead: tuple, p: int, device) -> torch.Tensor:
    """bool [B, P]: row r of subnet b is one of its leading lead[b] rows."""
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (lead, p, str(device), stream)       # per stream: made on another scene thread's stream it may not be written yet
    hit = _LEAD_PATTERNS.get(key)
    if hit is None:
        if len(_LEAD_PATTERNS) > 64:
            _LEAD_PATTERNS.clear()
        hit = torch.arange(p, device=device)[None, :] < torch.tensor(lead, device=device)[:, None]
        _LEAD_PATTERNS[key] = hit
    return hit


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        _xavier(self)

    def forward(self, x, query_pos=None):
        """Self-attention of the queries (blocks.py:9-44: q = k = x + query_pos, value = x, post-norm) written out: the
        projections, softmax(q k^T / sqrt(dh)) v and the output projection as plain fp32 matmuls.  `nn.MultiheadAttention`'s
        own forward dispatches to scaled_dot_product_attention, which on this ROCm build is an AOTriton flash kernel
        (`attn_fwd`): a Triton kernel on the path, and one whose fp32 products are not fp32-exact."""
        mha = self.self_attn
        # the written-out attention below has no dropout on the softmax weights (nn.MultiheadAttention applies it in training mode):
        # the reference builds these layers with dropout = 0.0 (transformer_predictor_v2.py:72-82); anything else must not pass silently
        assert mha.dropout == 0.0 or not self.training, "SelfAttentionLayer: attention dropout > 0 in training mode is not served"
        B, Q, D = x.shape
        H = mha.num_heads
        dh = D // H
        w, b = mha.in_proj_weight, mha.in_proj_bias
        qk = x if query_pos is None else x + query_pos
        qkp = F.linear(qk, w[:2 * D], b[:2 * D])
        q = qkp[..., :D].reshape(B, Q, H, dh).transpose(1, 2)
        k = qkp[..., D:].reshape(B, Q, H, dh).transpose(1, 2)
        v = F.linear(x, w[2 * D:], b[2 * D:]).reshape(B, Q, H, dh).transpose(1, 2)
        y = mha.out_proj(_attention_math(q, k, v).transpose(1, 2).reshape(B, Q, D))
        return self.norm(x + y)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        _xavier(self)

    def forward(self, q_embed, feats, attn_mask=None, pos=None, query_pos=None, mask_bits=None, feats_split=None,
                feats_shape=None):
        """q_embed [B,Q,D]; feats [B,N,D] (pass pos=None when the positional term is already added).  Mask
        either as `attn_mask` bool [B,Q,N] (True = masked, shared by the heads; torch path) or as `mask_bits`
        = (bits [B,N,4], any [B,4]) for the fused HIP kernel (ph_attn_cross_fwd), which also applies the
        all-masked -> unmasked rule."""
        q = self.norm(q_embed)
        mha = self.multihead_attn
        B, Q, D = q.shape
        H = self.nhead
        w, b = mha.in_proj_weight, mha.in_proj_bias
        qq = F.linear(q if query_pos is None else q + query_pos, w[:D], b[:D])
        if feats is None:      # the keys / values exist only as the pre-split operand of their projections
            assert pos is None and feats_split is not None and mask_bits is not None
            N = feats_shape[1]
            kv2, kv_split = None, feats_split
        else:
            kv = feats if pos is None else feats + pos
            N = kv.shape[1]
            kv2 = kv.reshape(B * N, D)
            kv_split = feats_split
            if kv_split is None and kv2.shape[0] >= fused_mod.MIN_ROWS_LINEAR:
                kv_split = split_rows_2d(kv2)                                     # one operand split for K and V
        kk = linear_rows(kv2, w[D:2 * D], b[D:2 * D], self, "k", in_split=kv_split).view(B, N, D)
        vv = linear_rows(kv2, w[2 * D:], b[2 * D:], self, "v", in_split=kv_split).view(B, N, D)
        qq = qq.view(B, Q, H, D // H).transpose(1, 2)
        if mask_bits is not None:
            be = backend_for(q.device)
            q4 = (qq * (float(D // H) ** -0.5)).contiguous()
            o = be.attn_cross_fwd(q4, kk.contiguous(), vv.contiguous(), mask_bits[0], mask_bits[1])
            return q + mha.out_proj(o)
        kk = kk.view(B, -1, H, D // H).transpose(1, 2)
        vv = vv.view(B, -1, H, D // H).transpose(1, 2)
        bias = None
        if attn_mask is not None:
            bias = torch.zeros(attn_mask.shape, dtype=q.dtype, device=q.device)
            bias.masked_fill_(attn_mask, float("-inf"))
            bias = bias[:, None]
        o = _attention_math(qq, kk, vv, bias)
        o = o.transpose(1, 2).reshape(B, Q, D)
        return q + mha.out_proj(o)


    def attend(self, q_embed, kk, vv, query_pos, mask_bits, split=None):
        """The layer with keys / values already projected ([B, N, D] each, or `split` = (K operand, V operand, N): the
        split f16 operands their projections emitted): query projection, masked attention (ph_attn_cross_fwd /
        ph_attn_cross_split), output projection, residual."""
        q = self.norm(q_embed)
        mha = self.multihead_attn
        B, Q, D = q.shape
        H = self.nhead
        w, b = mha.in_proj_weight, mha.in_proj_bias
        qq = F.linear(q if query_pos is None else q + query_pos, w[:D], b[:D]).view(B, Q, H, D // H).transpose(1, 2)
        be = backend_for(q.device)
        q4 = (qq * (float(D // H) ** -0.5)).contiguous()
        if split is not None:
            o = be.attn_cross_split(q4, split[0], split[1], split[2], mask_bits[0], mask_bits[1])
        else:
            o = be.attn_cross_fwd(q4, kk.contiguous(), vv.contiguous(), mask_bits[0], mask_bits[1])
        return q + mha.out_proj(o)

    def composed_kv(self, lin: nn.Linear, tab: torch.Tensor):
        """K and V projections composed with the level's input projection `lin` and with the position table:
            K = (x W_p^T + b_p + pos) W_k^T + b_k = x (W_k W_p)^T + (W_k b_p + b_k) + pos W_k^T,
        pos = [tab[x] | tab[y] | tab[z]]  =>  pos W_k^T = sum over the axes of (tab W_k[:, axis block]^T)[coordinate].
        -> dict(wk, bk, tk, wv, bv, tv): weights [D, cin], biases [D], tables [3, T, D]; cached per parameter version.
        (Products in fp64, rounded once: the composed map is the same linear map as the sequence, to fp32 rounding.)"""
        mha = self.multihead_attn
        w, b = mha.in_proj_weight, mha.in_proj_bias
        ver = tuple((p._version, p.data_ptr()) for p in (w, b, lin.weight, lin.bias)) + (tab.data_ptr(),)
        hit = self.__dict__.get("_ph_composed")
        if hit is not None and hit[0] == ver:
            return hit[1]
        D = w.shape[1]
        f = tab.shape[1]
        out = {}
        with torch.no_grad():
            wp, bp, t64 = lin.weight.double(), lin.bias.double(), tab.double()
            for name, sl in (("k", slice(D, 2 * D)), ("v", slice(2 * D, 3 * D))):
                wx, bx = w[sl].double(), b[sl].double()
                out["w" + name] = (wx @ wp).float().contiguous()
                out["b" + name] = (wx @ bp + bx).float().contiguous()
                out["t" + name] = torch.stack([t64 @ wx[:, a * f:(a + 1) * f].t() for a in range(3)]).float().contiguous()
        fused_mod.publish(w)
        self.__dict__["_ph_composed"] = (ver, out)
        return out


    def composed_feat(self, lin: nn.Linear, pe: "PositionEmbeddingSineSparse", exp2: int):
        """Everything `attend_feat` needs around the kernel, composed in fp64 and rounded once (cached per parameter version):
            w2 [H * (C + 16), D], b2: (q + query_pos) -> q2, i.e. query projection, 1/sqrt(dh) and
                                      per head [A_h^T | D0k | Gk * 2^-e | 0] in one linear map;
            mvo [H * (C + 16), D], co [D]: Y -> attention output, i.e. per head (B_h ; D0v ; Gv * 2^-e ; 0), the constants
                                      (bias + far position rows of V) and the output projection in one linear map.
        A = W_k W_p, B = W_v W_p (the level's input projection `lin` composed with the K / V projections); the position rows
        tab[t] W^T of an axis are tab_far W^T + [t == 0] D0 + eps_t G W^T (`PositionEmbeddingSineSparse.angle_model`)."""
        mha = self.multihead_attn
        w, b = mha.in_proj_weight, mha.in_proj_bias
        wo, bo = mha.out_proj.weight, mha.out_proj.bias
        tab = pe.table(w.device)
        ver = tuple((p._version, p.data_ptr()) for p in (w, b, wo, bo, lin.weight, lin.bias)) + (tab.data_ptr(), exp2)
        hit = self.__dict__.get("_ph_composed_feat")
        if hit is not None and hit[0] == ver:
            return hit[1]
        D = w.shape[1]
        H = self.nhead
        dh = D // H
        f = tab.shape[1]
        C = lin.weight.shape[1]
        E = C + 16
        _, G, i0, ifar = pe.angle_model(w.device)
        es = float(2.0 ** -pe.EPS_EXP2)
        with torch.no_grad():
            wp, bp, t64 = lin.weight.double(), lin.bias.double(), tab.double()
            wq, bq = w[:D].double(), b[:D].double()
            sides = {}
            for name, sl in (("k", slice(D, 2 * D)), ("v", slice(2 * D, 3 * D))):
                wx, bx = w[sl].double(), b[sl].double()
                A = wx @ wp                                                    # [D, C]
                d0 = torch.stack([(t64[i0] - t64[ifar]) @ wx[:, a * f:(a + 1) * f].t() for a in range(3)])     # [3, D]
                g = torch.stack([G @ wx[:, a * f:(a + 1) * f].t() for a in range(3)]) * es                      # [3, D]
                far = sum(t64[ifar] @ wx[:, a * f:(a + 1) * f].t() for a in range(3)) + wx @ bp + bx            # [D]
                M = torch.zeros((D, E), dtype=torch.float64, device=w.device)  # row = projection dim, column = key column
                M[:, :C] = A
                M[:, C:C + 3] = d0.t()
                M[:, C + 3:C + 6] = g.t()
                sides[name] = (M, far)
            Mk, _ = sides["k"]             # the keys' constant (bias + far rows) is the same for every key: softmax drops it
            Mv, cv = sides["v"]
            scale = float(dh) ** -0.5
            w2 = torch.zeros((H * E, D), dtype=torch.float64, device=w.device)
            b2 = torch.zeros((H * E,), dtype=torch.float64, device=w.device)
            mvo = torch.zeros((H * E, D), dtype=torch.float64, device=w.device)
            for h in range(H):
                hs = slice(h * dh, (h + 1) * dh)
                w2[h * E:(h + 1) * E] = (Mk[hs].t() @ wq[hs]) * scale         # [E, dh] @ [dh, D]
                b2[h * E:(h + 1) * E] = (Mk[hs].t() @ bq[hs]) * scale
                mvo[h * E:(h + 1) * E] = Mv[hs].t() @ wo.double()[:, hs].t()   # [E, dh] @ [dh, D]
            co = cv @ wo.double().t() + bo.double()
            out = dict(w2=w2.float().contiguous(), b2=b2.float().contiguous(), mvo=mvo.float().contiguous(),
                       co=co.float().contiguous(), E=E)
        fused_mod.publish(w)       # the cache serves every stream, the tensors were made on this one
        self.__dict__["_ph_composed_feat"] = (ver, out)
        return out

    def attend_feat(self, q_embed, comp, x_split, aug, n: int, query_pos, mask_bits):
        """The layer on the level's feature operand (ph_attn_cross_feat): keys and values are never formed."""
        q = self.norm(q_embed)
        B, Q, D = q.shape
        H, E = self.nhead, comp["E"]
        q2 = F.linear(q if query_pos is None else q + query_pos, comp["w2"], comp["b2"])
        q2 = q2.view(B, Q, H, E).transpose(1, 2).contiguous()
        be = backend_for(q.device)
        y = be.attn_cross_feat(q2, x_split, aug, n, mask_bits[0], mask_bits[1])            # [B, Q, H * E]
        return q + torch.addmm(comp["co"], y.view(B * Q, H * E), comp["mvo"]).view(B, Q, D)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        _xavier(self)

    def forward(self, x):
        x = self.norm(x)
        return x + self.linear2(F.relu(self.linear1(x)))


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.num_layers = num_layers
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < self.num_layers - 1:
                x = F.relu(x)
        return x


class LazyRows:
    """A sparse tensor whose feature rows are a row selection (index tensor, or an int n = the n leading rows) of a bigger matrix, gathered on first access
    (`.F` / `.C` / `.features` / `.coordinates`, or `.materialize()` for the ME.SparseTensor itself)."""

    def __init__(self, src: torch.Tensor, rows: torch.Tensor, key, mgr):
        self._src, self._rows = src, rows
        self.coordinate_map_key, self.coordinate_manager = key, mgr
        self._st = None

    def materialize(self):
        if self._st is None:
            rows = self._src[:self._rows] if isinstance(self._rows, int) else self._src.index_select(0, self._rows)
            self._st = ME.SparseTensor(rows, coordinate_map_key=self.coordinate_map_key,
                                       coordinate_manager=self.coordinate_manager)
            self._src = self._rows = None
        return self._st

    def __getattr__(self, name):          # F, C, features, coordinates, dense(), ...: whatever SparseTensor offers
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


_PREDICTORS = None       # weak set of live predictors: `release_stream_graphs` walks it


def release_stream_graphs(stream) -> int:
    """Drop the query-side hipGraphs every live predictor captured for `stream` (a torch.cuda.Stream or a raw handle): they
    are keyed by the launch stream, so a serving loop that retires its streams (SceneServer.close) would otherwise leave up
    to _QGRAPH_MAX captures (graph + static buffers) behind per predictor.  Returns the number dropped."""
    handle = int(getattr(stream, "cuda_stream", stream) or 0)
    n = 0
    for tp in list(_PREDICTORS or ()):
        graphs = tp.__dict__.get("_qgraphs")
        if graphs:
            for k in [k for k in graphs if k[-1] == handle]:
                del graphs[k]
                n += 1
    return n


class TransformerPredictorV2(nn.Module):
    def __init__(self, in_channels, num_classes=20, hidden_dim=384, num_queries=100, nheads=8,
                 dim_feedforward=2048, mask_dim=256, n_infers=2, **_ignored):
        super().__init__()
        global _PREDICTORS
        if _PREDICTORS is None:
            import weakref
            _PREDICTORS = weakref.WeakSet()
        _PREDICTORS.add(self)
        self.nheads = nheads
        self.n_infers = n_infers
        self.hidden_dim = hidden_dim
        self.query_dim = hidden_dim
        self.num_queries = num_queries
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.pe_layer = PositionEmbeddingSineSparse(hidden_dim // 3, normalize=True)
        self.src_scales = [4, 2, 1]
        self.num_layers = len(self.src_scales)
        self.transformer_self_attention_layers = nn.ModuleList(
            SelfAttentionLayer(hidden_dim, nheads) for _ in range(self.num_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(
            CrossAttentionLayer(hidden_dim, nheads) for _ in range(self.num_layers))
        self.transformer_ffn_layers = nn.ModuleList(
            FFNLayer(hidden_dim, dim_feedforward) for _ in range(self.num_layers))
        self.query_feat = nn.Embedding(num_queries * n_infers, hidden_dim)
        self.query_embed = nn.Embedding(num_queries * n_infers, hidden_dim)
        self.input_projs = nn.ModuleList(nn.Linear(in_channels[i], hidden_dim) for i in range(self.num_layers))
        self.max_pools = nn.ModuleDict({
            str(s): ME.MinkowskiMaxPooling(kernel_size=s, stride=s, dimension=3) for s in self.src_scales})
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, hidden_dim, 3)
        self.mask_feat_proj = nn.Linear(mask_dim, hidden_dim)

    # -- heads --------------------------------------------------------------------------------------
    def heads_query_side(self, output, want_operand):
        """The part of `pred_heads` that only touches the [B, Q, D] queries: class logits, mask embedding and -
        for the split kernel - the mask embedding in operand form.
        want_operand == 2: the ABSORBED form.  The voxel features of the mask heads are `x W_p^T + b_p + pos` and are
        read only through `voxel_feat . e_q`, so
            mask[n, q] = x[n] . (W_p^T e_q) + b_p . e_q + tab[x_n] . e_q[block 0] + tab[y_n] . e_q[block 1] + tab[z_n] . e_q[block 2]
        i.e. a C -> Q product on the level's own features with a per-subnet bias and three table rows per voxel
        (ph_conv_desc.axis_table): the [N, D] voxel features are never formed.  Returned: (w_c operand, unscale,
        bias [B, Q], tables [B, 3, T, Q])."""
        d = self.decoder_norm(output)
        outputs_class = self.class_embed(d)
        mask_embed = self.mask_embed(d)                                   # [B,Q,D]
        if want_operand == 2:
            lin = self.mask_feat_proj
            B, Q, D = mask_embed.shape
            wc = torch.matmul(mask_embed, lin.weight)                      # [B, Q, C]
            bc = torch.matmul(mask_embed, lin.bias)                        # [B, Q]
            tab = self.pe_layer.table(mask_embed.device)                   # [T, f]
            f = tab.shape[1]
            tq = torch.matmul(tab, mask_embed.view(B, Q, 3, f).permute(0, 2, 3, 1))      # [B, 3, T, Q]
            return outputs_class, mask_embed, prepare_batched_weights(wc) + (bc.contiguous(), tq.contiguous())
        prepared = prepare_batched_weights(mask_embed) if want_operand else None
        return outputs_class, mask_embed, prepared

    def pred_heads(self, output, mask_features, mask_features_split=None, shape=None, query_side=None, absorbed=None):
        """`absorbed` = dict(x_split, coords [B, P, 4], shape (B, P, C)): the heads read the level's features directly
        (heads_query_side with want_operand == 2)."""
        if absorbed is not None:
            outputs_class, _, prep = query_side if query_side is not None else self.heads_query_side(output, 2)
            w_split, unscale, bc, tq = prep
            outputs_mask = batched_rows_matmul(None, None, absorbed["x_split"], shape=absorbed["shape"],
                                               prepared=(w_split, unscale), bias=bc,
                                               axis=(tq, absorbed["coords"], self.pe_layer.TABLE_LO))
            return outputs_class, outputs_mask
        outputs_class, mask_embed, prepared = query_side if query_side is not None else \
            self.heads_query_side(output, mask_features_split is not None)
        if mask_features_split is not None:
            outputs_mask = batched_rows_matmul(mask_features, mask_embed, mask_features_split, shape=shape,
                                               prepared=prepared)
        else:
            outputs_mask = torch.matmul(mask_features, mask_embed.transpose(1, 2))   # [B,P,Q]
        return outputs_class, outputs_mask

    # -- fixed-shape query-side ops as one replayed graph -------------------------------------------------------
    def _query_step(self, layer: int, output, query_embed, want_operand: bool):
        """self-attention + FFN of decoder layer `layer` (layer < 0: none) followed by the query side of the heads.
        Everything here has the static shape [B, Q, D] and launches ~40 tiny kernels: on the GPU it is captured
        once per (layer, shape) into a hipGraph and replayed (`PASCO_QUERY_GRAPH=0` or any capture failure ->
        eager)."""
        if layer >= 0:
            output = self.transformer_self_attention_layers[layer](output, query_pos=query_embed)
            output = self.transformer_ffn_layers[layer](output)
        return (output,) + tuple(self.heads_query_side(output, want_operand))

    def query_graph_state(self) -> str:
        """"graph" (query-side ops replayed from captured hipGraphs), "eager" (switched off: PASCO_QUERY_GRAPH=0, CPU,
        training) or "eager (capture failed: ...)" - a capture failure is a performance cliff, so serving loops and
        bench.py report it instead of leaving a one-off warning behind."""
        broken = self.__dict__.get("_qgraph_broken", False)
        if broken:
            return f"eager (capture failed: {broken})"
        if os.environ.get("PASCO_QUERY_GRAPH", "1") == "0" or not self.__dict__.get("_qgraphs"):
            return "eager"
        return "graph"

    def param_versions(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    _QGRAPH_MAX = 48     # captured graphs kept per module (4 per shape and stream; a few shapes, a few streams); oldest evicted

    def query_step(self, layer: int, output, query_embed, want_operand: bool, vers=None):
        if not output.is_cuda or self.training or torch.is_grad_enabled() or \
                os.environ.get("PASCO_QUERY_GRAPH", "1") == "0" or self.__dict__.get("_qgraph_broken", False):
            return self._query_step(layer, output, query_embed, want_operand)
        graphs = self.__dict__.setdefault("_qgraphs", {})
        # a captured graph bakes in the ADDRESSES of every parameter it reads: the key carries version and storage
        # address of each (load_state_dict(assign=True), .cpu().cuda() round trips, param.data = ... change the
        # address without bumping the version)
        # (`vers` from the caller: forward() takes it once for its four calls - walking the module tree costs ~0.2 ms)
        if vers is None:
            vers = self.param_versions()
        # one graph per launch stream: a serving loop with several scenes in flight replays them concurrently, and a
        # graph's static input / output buffers must not be shared between streams
        key = (layer, tuple(output.shape), tuple(query_embed.shape), output.device, want_operand,
               torch.cuda.current_stream(output.device).cuda_stream)
        hit = graphs.get(key)
        if hit is None or hit["vers"] != vers:
            try:
                with _CAPTURE_LOCK:
                    hit = self._capture_query_step(layer, output, query_embed, want_operand)
                hit["vers"] = vers
                graphs.pop(key, None)
                graphs[key] = hit
                while len(graphs) > self._QGRAPH_MAX:          # dicts keep insertion order: drop the oldest capture
                    graphs.pop(next(iter(graphs)))
            except Exception as exc:       # capture is an optimisation: never let it take the step down
                self.__dict__["_qgraph_broken"] = f"{type(exc).__name__}: {exc}"
                import warnings
                warnings.warn(f"pasco_amd: query-side graph capture failed ({type(exc).__name__}: {exc}); running eagerly")
                return self._query_step(layer, output, query_embed, want_operand)
        hit["x"].copy_(output)
        hit["qe"].copy_(query_embed)       # per-call tensor (a row selection of the embedding with subnets): static copy
        hit["graph"].replay()
        out, oc, me, prepared = hit["outs"]
        # the class logits are kept by the caller across replays -> private copy; the rest is consumed before the next replay
        return out, oc.clone(), me, prepared

    def _capture_query_step(self, layer, output, query_embed, want_operand):
        x = output.detach().clone()
        qe = query_embed.detach().clone()
        be = backend_for(output.device)
        # the captured launches report into the status word of the stream the graph will be replayed on (this one)
        with be.pin_status(output.device):
            side = torch.cuda.Stream(device=output.device)
            side.wait_stream(torch.cuda.current_stream(output.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):             # warm-up outside the capture: library handles, workspaces, autotuning
                    self._query_step(layer, x, qe, want_operand)
            torch.cuda.current_stream(output.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads may be launching
                outs = self._query_step(layer, x, qe, want_operand)
        return {"graph": g, "x": x, "qe": qe, "outs": outs}

    # -- attention mask -----------------------------------------------------------------------------
    def compute_mask_bits(self, outputs_mask, voxel_coord, src_C, src_scale, min_Cs, max_Cs, cache=None):
        """Attention mask of one level as bits: (bits int32 [B, N_level, 4], any int32 [B, 4]).

        Query q may attend level voxel p iff some scale-1 voxel v of the same subnet inside p's
        s^3 block has mask_logit[v, q] > 0 (sigmoid > 0.5) (transformer_predictor_v2.py:220-289).
        The reference builds a 0/1 float SparseTensor, max-pools it, densifies it to [1,Q,X,Y,Z] and
        indexes it; here the mask is 1 bit per (voxel, query) from the start: pack, OR-pool over the
        children, hash lookup of the level voxels' sites - same values, 25x less traffic."""
        B, P, Q = outputs_mask.shape
        dev = src_C.device
        be = backend_for(dev)
        bits1, _ = be.attn_mask_pack(outputs_mask.reshape(B * P, Q).contiguous(), B, P, positive_only=True,
                                     want_any=False)
        if cache is not None and "key1" in cache:            # the scale-1 map depends on the coordinates only
            mgr, key1, uniq = cache["key1"]
        else:
            bcol = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(P).reshape(-1, 1)
            keep_C = torch.cat([bcol, voxel_coord.reshape(B * P, 4)[:, 1:].to(torch.int32)], dim=1).contiguous()
            mgr = ME.CoordinateManager(D=3, device=dev)
            key1, (_, uniq) = mgr.insert_and_map(keep_C, 1)    # duplicated (padded) rows keep their first occurrence
            if cache is not None:
                cache["key1"] = (mgr, key1, uniq)
        bits1 = bits1.reshape(B * P, 4)
        if uniq is not None:
            bits1 = be.gather_rows(bits1, uniq)
        N = src_C.shape[1]
        # Fast path: the level voxel's bits = OR over the fine voxels of its s^3 block, straight from the fine map's hash table
        # (ph_bits_block_or) - no pooled map, no neighbour table, no dense-site map.  Exactly the reference's max-pool ->
        # dense -> index whenever no coordinate lies outside the subnet's [min, max] box (then the reference's dense
        # indexing wraps negative indices: the path below reproduces that); the test is made on the device and read once.
        if be.has("bits_block_or") and os.environ.get("PASCO_MASK_BLOCK", "1") != "0":
            fine = mgr._maps[key1]
            # Host-side decision where the host knows the boxes (UNet3DV2.forward read them once): every REAL voxel of the
            # fine map and of the level lies inside its subnet's box (the panoptic branch prunes with exactly that test,
            # decoder_v3.py:151-158,415-420), so only PADDED rows - coordinate (0, 0, 0) - can lie outside, and they do iff
            # the subnet has padding and 0 is outside [min, max] on some axis.  No device read, no speculation.
            hmin, hmax = getattr(min_Cs, "host", None), getattr(max_Cs, "host", None)
            lens = cache.get("lens") if cache is not None else None
            if hmin is not None and hmax is not None and lens is not None and len(hmin) == B:
                pad_out = any(lens["fine"][b] < P or lens["level"][src_scale][b] < N for b in range(B)
                              if any(lo > 0 for lo in hmin[b]) or any(hi < 0 for hi in hmax[b]))
                if not pad_out:
                    out = be.bits_block_or(src_C.reshape(B * N, 4).to(torch.int32).contiguous(), N, src_scale, fine.tkeys,
                                           fine.tvals, bits1.contiguous())
                    bits = out.reshape(B, N, 4)
                    return bits, be.bits_or_reduce(bits)
                # a padded row outside its box: the exact dense-index path below (python-style wrap of negative indices)
            else:       # boxes only on the device: test there, one read
                if cache is not None and "bounds" in cache:
                    mn32, mx32, fine_bad = cache["bounds"]
                else:
                    mn32 = torch.stack([torch.as_tensor(m) for m in min_Cs]).to(dev).to(torch.int32).contiguous()    # [B, 3]
                    mx32 = torch.stack([torch.as_tensor(m) for m in max_Cs]).to(dev).to(torch.int32).contiguous()
                    vc = voxel_coord.reshape(B, P, 4)[..., 1:].to(torch.int32)
                    fine_bad = ((vc < mn32[:, None]) | (vc > mx32[:, None])).any()
                    if cache is not None:
                        cache["bounds"] = (mn32, mx32, fine_bad)
                out, rng = be.bits_block_or(src_C.reshape(B * N, 4).to(torch.int32).contiguous(), N, src_scale, fine.tkeys,
                                            fine.tvals, bits1.contiguous(), mn32, mx32, want_range=True)
                if not bool(((rng != 0) | fine_bad).item()):
                    bits = out.reshape(B, N, 4)
                    return bits, be.bits_or_reduce(bits)
        if src_scale != 1:
            pool = self.max_pools[str(src_scale)]
            keyp = mgr.stride(key1, pool.stride)
            nbr = mgr.kernel_map(key1, keyp, pool.kernel_size, pool.dilation)
            pooled_bits = be.bits_orpool(bits1.contiguous(), nbr)
        else:
            keyp, pooled_bits = key1, bits1
        mn = torch.stack([torch.as_tensor(m) for m in min_Cs]).to(dev).to(torch.int64)   # [B,3]
        mx = torch.stack([torch.as_tensor(m) for m in max_Cs]).to(dev).to(torch.int64)
        size = torch.div(mx - mn, src_scale, rounding_mode="floor") + 1                   # dense extent per subnet

        def sites(coords4, b_index):
            """dense-grid site of each coordinate, with python-style wrap of negative indices (the
            reference indexes a dense tensor: SparseTensor.dense + advanced indexing)."""
            idx = torch.div(coords4[:, 1:].to(torch.int64) - mn[b_index], src_scale, rounding_mode="floor")
            idx = torch.where(idx < 0, idx + size[b_index], idx)
            return torch.cat([b_index.reshape(-1, 1), idx], dim=1).to(torch.int32)

        pc = mgr.get_coordinates(keyp)
        site_mgr = ME.CoordinateManager(D=3, device=dev)
        site_key, (_, suniq) = site_mgr.insert_and_map(sites(pc, pc[:, 0].to(torch.int64)), 1)
        bq = torch.arange(B, device=dev, dtype=torch.int64).repeat_interleave(N)
        rows = site_mgr.find(site_key, sites(src_C.reshape(B * N, 4), bq))
        if suniq is not None:   # two pooled voxels wrapped onto one site: keep the first
            rows = torch.where(rows >= 0, suniq[rows.clamp(min=0).long()], rows)
        bits = be.gather_rows(pooled_bits.contiguous(), rows.contiguous()).reshape(B, N, 4)   # -1 -> no bit
        return bits, be.bits_or_reduce(bits)

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, xs, sem_logits, min_Cs, max_Cs, keep_pad, subnets=None, sem_tensors=None):
        """xs[scale] = (feats [B,N,C], coords [B,N,4]); returns one dict per subnet.
        `sem_tensors` (optional): the per-subnet sparse tensors the padded batch rows were made of (same rows, same
        order) - when every row of a subnet is kept its voxel logits reuse that tensor's coordinate map instead of
        hashing the same coordinates again.
        `subnets` (optional list of subnet indices): the batch rows hold only those subnets' voxels and
        only their query sets run (subnet-parallel heads)."""
        sem_F, sem_C = sem_logits
        B = sem_F.shape[0]
        D = self.hidden_dim
        output = self.query_feat.weight.reshape(self.n_infers, -1, D)
        query_embed = self.query_embed.weight.reshape(self.n_infers, -1, D)
        if subnets is None:
            assert B == self.n_infers, "batch size should be equal to number of inference"
        else:
            assert B == len(subnets)
            sel = torch.as_tensor(list(subnets), device=output.device)
            output, query_embed = output[sel], query_embed[sel]
            min_Cs = [min_Cs[i] for i in subnets]
            max_Cs = [max_Cs[i] for i in subnets]
        srcs, src_Cs = [], []
        for s in self.src_scales:
            f, c = xs[s]
            srcs.append(f)
            src_Cs.append(c)
        pos_cache = {}

        def pos_of(i):          # materialised only on the paths that need the encoding as a tensor
            if i not in pos_cache:
                pos_cache[i] = self.pe_layer(src_Cs[i].reshape(-1, 4), coff=1).reshape(B, -1, D)
            return pos_cache[i]

        voxel_coord = xs[1][1]
        x1 = xs[1][0]
        dev = x1.device
        # with the convolution kernel the position encoding is never materialised: it enters the projections' epilogues
        # as three table rows per voxel (ph_conv_desc.axis_table)
        use_tables = fused_mod.fusion() and fused_mod._kernel_device(dev) and fused_mod.conv_precision() == "f16x3" and \
            os.environ.get("PASCO_PE_TABLE", "1") != "0"
        tab = self.pe_layer.table(dev) if use_tables else None
        # voxel features of the mask heads: read only as the operand of `voxel_feat @ mask_embed^T`, so with the
        # split kernel the projection writes that operand directly (no fp32 copy, no separate split pass)
        P = x1.shape[1]
        heads_split = self.num_queries % 4 == 0
        c1 = src_Cs[-1].reshape(-1, 4)
        tables_ok = use_tables and c1.dtype == torch.int32 and x1.shape[0] * P >= fused_mod.MIN_ROWS_LINEAR
        # absorbed mask heads (heads_query_side): no [N, D] voxel features at all
        absorbed = None
        if tables_ok and heads_split and x1.shape[-1] % 8 == 0 and os.environ.get("PASCO_HEAD_ABSORB", "1") != "0":
            x1_split = split_rows_2d(x1.reshape(-1, x1.shape[-1]))
            if x1_split is not None:
                absorbed = dict(x_split=x1_split, coords=c1.contiguous().view(B, P, 4), shape=(B, P, x1.shape[-1]))
        if absorbed is not None:
            voxel_feat, vf_split = None, None
        elif tables_ok:
            voxel_feat, vf_split = linear_rows(x1.reshape(-1, x1.shape[-1]), self.mask_feat_proj.weight,
                                               self.mask_feat_proj.bias, self.mask_feat_proj, "w",
                                               axis=(self.pe_layer.block_table(dev), c1.contiguous(), self.pe_layer.TABLE_LO),
                                               emit=True, want_out=not heads_split)
        else:
            voxel_feat, vf_split = linear_rows(x1.reshape(-1, x1.shape[-1]), self.mask_feat_proj.weight,
                                               self.mask_feat_proj.bias, self.mask_feat_proj, "w",
                                               residual=pos_of(len(self.src_scales) - 1).reshape(-1, D), emit=True,
                                               want_out=not heads_split)
        if not heads_split:
            vf_split = None
        if voxel_feat is not None:
            voxel_feat = voxel_feat.view(B, -1, D)
        vf_shape = (B, P, D)
        predictions_class, predictions_mask = [], []
        mask_cache = {}
        if "lens" in xs.get("_meta", {}):       # rows of every subnet at every scale (host ints from the caller)
            mask_cache["lens"] = xs["_meta"]["lens"]
        head_mode = 2 if absorbed is not None else (vf_split is not None)
        qvers = self.param_versions() if output.is_cuda else None
        output, *qs = self.query_step(-1, output.contiguous(), query_embed, head_mode, qvers)
        oc, om = self.pred_heads(output, voxel_feat, vf_split, vf_shape, query_side=qs, absorbed=absorbed)
        predictions_class.append(oc)
        predictions_mask.append(om)
        for i in range(self.num_layers):
            lin = self.input_projs[i]
            # input projection with the positional term added in the same launch: the layer only ever uses
            # src + pos (transformer/blocks.py:83-86, key = value = bb_feat + pos)
            N_i, Qn = srcs[i].shape[1], om.shape[2]
            be = backend_for(srcs[i].device)
            fused_attn = be.attn_supported(Qn, D // self.nheads)
            ca = self.transformer_cross_attention_layers[i]
            ci = src_Cs[i].reshape(-1, 4)
            bits, any_ = self.compute_mask_bits(om, voxel_coord, src_Cs[i], self.src_scales[i], min_Cs, max_Cs,
                                                cache=mask_cache)
            tall = fused_attn and use_tables and ci.dtype == torch.int32 and B * N_i >= fused_mod.MIN_ROWS_LINEAR
            x2 = x_split = None
            if tall:
                x2 = srcs[i].reshape(-1, srcs[i].shape[-1])
                # the finest level's features are the mask heads' operand too: split once
                x_split = absorbed["x_split"] if (absorbed is not None and srcs[i] is x1) else split_rows_2d(x2)
                ci = ci.contiguous()
            if tall and x_split is not None and be.attn_feat_supported(Qn, x2.shape[1]) and \
                    os.environ.get("PASCO_ATTN_FEAT", "1") != "0":
                # attention straight on the level's feature operand: K and V (two [N, 384] operands written once and read
                # once) and their projection launches do not exist (CrossAttentionLayer.composed_feat, ph_attn_cross_feat)
                from ..me.backend import SPLIT_ACT_EXP2
                comp = ca.composed_feat(lin, self.pe_layer, SPLIT_ACT_EXP2)
                eps = self.pe_layer.angle_model(dev)[0]
                aug = be.pos_aug(ci, eps, self.pe_layer.TABLE_LO)
                output = ca.attend_feat(output, comp, x_split, aug, N_i, query_embed, (bits, any_))
            elif tall:
                # K and V straight from the level's features: input projection, position term and K / V projection
                # composed into one launch each (CrossAttentionLayer.composed_kv)
                cm = ca.composed_kv(lin, tab)
                # ... written only as split f16 operands, which the attention kernel streams (ph_attn_cross_split)
                kk, k_op = linear_rows(x2, cm["wk"], cm["bk"], ca, "ck", in_split=x_split, emit=True, want_out=False,
                                       axis=(cm["tk"], ci, self.pe_layer.TABLE_LO))
                vv, v_op = linear_rows(x2, cm["wv"], cm["bv"], ca, "cv", in_split=x_split, emit=True, want_out=False,
                                       axis=(cm["tv"], ci, self.pe_layer.TABLE_LO))
                if k_op is not None and v_op is not None and os.environ.get("PASCO_ATTN_SPLIT", "1") != "0":
                    output = ca.attend(output, None, None, query_embed, (bits, any_), split=(k_op, v_op, N_i))
                else:
                    kk = kk if kk is not None else _unsplit_rows(k_op, D)
                    vv = vv if vv is not None else _unsplit_rows(v_op, D)
                    output = ca.attend(output, kk.view(B, N_i, D), vv.view(B, N_i, D), query_embed, (bits, any_))
            elif fused_attn:
                # with the fused attention kernel src + pos is read only by the K / V projections: emit it as their
                # operand (no fp32 copy)
                src_F, src_split = linear_rows(srcs[i].reshape(-1, srcs[i].shape[-1]), lin.weight, lin.bias, lin, "w",
                                               residual=pos_of(i).reshape(-1, D), emit=True, want_out=False)
                if src_F is not None:
                    src_F = src_F.view(B, -1, D)
                output = ca(output, src_F, pos=None, query_pos=query_embed, mask_bits=(bits, any_),
                            feats_split=src_split, feats_shape=(B, N_i, D))
            else:   # shapes outside the fused kernel: torch attention with the materialised bool mask
                src_F = linear_rows(srcs[i].reshape(-1, srcs[i].shape[-1]), lin.weight, lin.bias, lin, "w",
                                    residual=pos_of(i).reshape(-1, D)).view(B, -1, D)
                q_idx = torch.arange(Qn, device=bits.device)
                allow = (bits[:, :, (q_idx >> 5).long()] >> (q_idx & 31).to(torch.int32)) & 1      # [B,N,Q]
                attn_mask = ~(allow != 0).permute(0, 2, 1)
                attn_mask = attn_mask & ~attn_mask.all(dim=-1, keepdim=True)   # all-masked -> unmasked
                output = ca(output, src_F, attn_mask=attn_mask, pos=None, query_pos=query_embed)
            output, *qs = self.query_step(i, output.contiguous(), query_embed, head_mode, qvers)
            oc, om = self.pred_heads(output, voxel_feat, vf_split, vf_shape, query_side=qs, absorbed=absorbed)
            predictions_class.append(oc)
            predictions_mask.append(om)
        panop_predictions = []
        # Optimistic row selection (no host read): the kept rows of subnet b are normally exactly its n_b leading rows (the
        # batch was padded behind them).  That is verified ON THE DEVICE - one comparison of keep_pad with the expected
        # pattern, folded into the stream's optimistic word - and the outputs are row slices; should it not hold, the
        # end-of-step check redoes the step through the compaction below.
        lead = None
        opt = fused_mod.optimistic_word(keep_pad.device) if sem_tensors is not None else None
        if opt is not None and all(t is not None and t.F.shape[0] <= voxel_coord.shape[1] for t in sem_tensors):
            lead = [int(t.F.shape[0]) for t in sem_tensors]
            # leading n_b rows kept, nothing behind them: ONE comparison with the expected pattern (a function of the row counts -
            # host ints - kept per (counts, pad length, device, stream)) instead of four small launches per subnet
            bad = (keep_pad != _lead_pattern(tuple(lead), keep_pad.shape[1], keep_pad.device)).any()
            opt.bitwise_or_(bad.to(torch.int32))
        for b in range(B):
            if lead is not None:
                src, n_b = sem_tensors[b], lead[b]
                key, mgr = src.coordinate_map_key, src.coordinate_manager
                first = ME.SparseTensor(predictions_mask[0][b][:n_b], coordinate_map_key=key, coordinate_manager=mgr)
                aux_masks = [first] + [LazyRows(m[b], n_b, key, mgr) for m in predictions_mask[1:-1]]
                last = ME.SparseTensor(predictions_mask[-1][b][:n_b], coordinate_map_key=key,
                                       coordinate_manager=mgr) if len(predictions_mask) > 1 else first
            else:
                kept = keep_pad[b].nonzero().reshape(-1)          # one compaction per subnet, reused by every mask
                src = sem_tensors[b] if sem_tensors is not None else None
                if src is not None and kept.shape[0] == src.F.shape[0] and src.F.shape[0] <= voxel_coord.shape[1]:
                    # kept = 0 .. n-1 (ascending, distinct, all below n): the rows ARE the source tensor's rows
                    first = ME.SparseTensor(predictions_mask[0][b].index_select(0, kept),
                                            coordinate_map_key=src.coordinate_map_key, coordinate_manager=src.coordinate_manager)
                else:
                    first = ME.SparseTensor(predictions_mask[0][b].index_select(0, kept), voxel_coord[b].index_select(0, kept))
                key, mgr = first.coordinate_map_key, first.coordinate_manager
                idx = first.unique_index        # None unless coordinates repeat
                rows = kept if idx is None else kept.index_select(0, idx.long())
                # Only the LAST prediction's voxel logits feed the inference path (ensembling, panoptic_inference); the
                # auxiliary ones exist for the training loss (net_panoptic_sparse.py:437-451).  Each is an 84 MB row gather
                # at S10, so they are gathered when somebody asks for them, not per step.
                aux_masks = [first] + [LazyRows(m[b], rows, key, mgr) for m in predictions_mask[1:-1]]
                last = ME.SparseTensor(predictions_mask[-1][b].index_select(0, rows), coordinate_map_key=key,
                                       coordinate_manager=mgr) if len(predictions_mask) > 1 else first
            classes = [c[b].unsqueeze(0) for c in predictions_class]
            panop_predictions.append({
                "query_logits": classes[-1],
                "voxel_logits": last,
                "aux_outputs": [{"query_logits": a, "voxel_logits": m} for a, m in zip(classes[:-1], aux_masks)],
            })
        return panop_predictions
