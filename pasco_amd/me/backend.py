"""ctypes binding of the flat C ABI declared in include/pasco_hip.h.

The product backend is libpascohip.so (hand-written HIP for gfx950).  There is NO CPU fallback in
this package: asking for a backend for a CPU tensor raises unless a test harness has explicitly
registered a checker library (tests register oracle/libpasco_oracle.so, which exports the same ABI
under the `pho_` prefix).  On a machine with a GPU a missing/unloadable libpascohip.so is a hard
error.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libpascohip.so")

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
ABI_VERSION = 5          # include/pasco_hip.h PH_ABI_VERSION this binding was written against


class StatusError(RuntimeError):
    """A device-side status flag was raised since the last check (`CBackend.check_status`).  `bits` holds every flag:
    1 = f16 range of a split-precision operand, 2 = unpackable coordinate, 4 = coordinate outside a per-axis table,
    8 = the fused input stage must be redone on its general path (all-zero merged row), 16 = an optimistic shortcut of the
    graph did not hold (redo with the checked paths), 32 = a kernel map handed over as one-pair-per-row (row lists) has
    another number of pairs than rows."""

    def __init__(self, bits: int, message: str):
        super().__init__(message)
        self.bits = bits


class F16RangeError(StatusError):
    """ONLY bit 0 was raised: an activation left the f16 range of the split-precision convolutions.  The result of the
    step is unusable, the exact-fp32 path (fused.set_conv_precision("f32")) serves the same graph."""
MAX_KVOL = 64
# Activation operands of the split-precision convolutions stand for x * 2^SPLIT_ACT_EXP2 (include/pasco_hip.h
# `split_exp2`): an f16 lo half is normal only while |x| * 2^e >= 2^-3, so unscaled activations below 0.125 lost
# operand bits (absolute floor 2^-25).  With e = 5 the full 22-bit operand covers |x| in [2^-8, 2047] and the range
# flag fires above 2047.
SPLIT_ACT_EXP2 = 5

_vp = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32


class ConvDesc(C.Structure):
    """Mirror of `ph_conv_desc` (include/pasco_hip.h)."""

    _fields_ = [
        ("in_", _vp), ("weight", _vp), ("nbr", _vp), ("out", _vp),
        ("n_in", _i64), ("n_out", _i64),
        ("cin", _i32), ("cout", _i32), ("kvol", _i32), ("pro_act", _i32),
        ("pro_scale", _vp), ("pro_shift", _vp), ("bias", _vp),
        ("epi_scale", _vp), ("epi_shift", _vp),
        ("epi_act", _i32), ("epi_slope", C.c_float),
        ("residual", _vp), ("res_act", _i32), ("split_exp2", _i32),
        ("epi2_scale", _vp), ("epi2_shift", _vp),
        ("mma_mode", _i32), ("w_unscale", C.c_float), ("w_f16_hi", _vp), ("w_f16_lo", _vp),
        ("splitk_ws", _vp), ("splitk_ws_bytes", _i64), ("status", _vp),
        ("in_split", _vp), ("w_split", _vp),
        ("out_split", _vp), ("osp_scale", _vp), ("osp_shift", _vp), ("osp_act", _i32), ("route", _i32),
        ("win_rows", _vp), ("win_cnt", _vp), ("win_slots", _vp), ("win_stats", _vp),
        ("axis_table", _vp), ("axis_coords", _vp), ("axis_lo", _i32), ("axis_rows", _i32),
        ("rl_in", _vp), ("rl_out", _vp), ("rl_tile_k", _vp), ("rl_rows", _i64), ("rl_tiles", _i32),
        ("exact_if", _vp), ("w_frag", _vp),
        ("grid_dims", _i32 * 4), ("grid_kernel", _i32 * 3), ("reserved3", _i32),
    ]


class SemEnsDesc(C.Structure):
    """Mirror of `ph_sem_ens_desc` (include/pasco_hip.h)."""

    _fields_ = [("m", _i32), ("c", _i32), ("n_sites", _i64), ("logits", _vp * 8), ("rows", _vp * 8),
                ("out", _vp * 9), ("conf", _vp * 9)]


# name -> argtypes (everything returns int unless listed in _RESTYPES)
_SIGNATURES = {
    "abi_version": [],
    "conv_desc_size": [],
    "last_error": [],
    "workspace_bytes": [_i64],
    "map_insert": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "map_find": [_vp, _i64, _vp, _vp, _i64, _vp, _vp],
    "coords_floor": [_vp, _i64, _i32, _vp, _vp],
    "coords_expand": [_vp, _i64, _i32, _vp, _vp],
    "nbr_build": [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _vp, _vp],
    "nbr_build_same": [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _vp, _vp],
    "kmap_compact": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _i64, _vp],
    "rowlist_pack": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "conv_fwd": [C.POINTER(ConvDesc), _vp],
    "conv_last_config": [_vp],
    "win_build": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp],
    "sem_ensemble": [C.POINTER(SemEnsDesc), _vp],
    "split_rows": [_vp, _i64, _i32, _vp, _vp, _i32, C.c_float, _i32, _vp, _vp, _vp],
    "maxpool_fwd": [_vp, _i32, _vp, _i32, _i64, _vp, _vp],
    "mask_compact": [_vp, _i64, _vp, _vp, _vp, _i64, _vp],
    "gather_rows": [_vp, _i32, _vp, _i64, _vp, _vp],
    "scatter_add_rows": [_vp, _i32, _vp, _i64, _vp, _vp],
    "to_dense": [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp],
    "to_sparse_coords": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "dense_gather": [_vp, _i32, _vp, _vp, _i64, _vp, _vp],
    "sine_pe": [_vp, _i64, _i32, _i32, _i32, _vp, C.c_float, _vp, _i32, _i32, _vp, _vp],
    "attn_workspace_bytes": [_i64, _i32, _i32, _i32, _i32],
    "attn_mask_pack": [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "bits_orpool": [_vp, _vp, _i32, _i64, _vp, _vp],
    "bits_or_reduce": [_vp, _i64, _i32, _vp, _vp],
    "attn_cross_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "bits_block_or": [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "project_canonical": [_vp, _i32, _i32, _i32, C.c_double, _vp, _vp, _vp],
    "ens_resample": [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp],
    "ens_merge": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "ens_finish": [_vp, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp],
    "attn_cross_split": [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp],
    "attn_cross_feat": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp],
    "pos_aug": [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "keep_mask": [_vp, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp],
    "points_bounds": [_vp, _i64, _vp, _vp],
    "points_mark": [_vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "mask_compact_rank": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp],
    "points_link": [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "cells_max": [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "panop_queries": [_vp, _i32, _i32, C.c_float, _vp, _vp, _vp],
    "panop_argmax": [_vp, _i64, _i32, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp],
    "panop_write": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
}
_RESTYPES = {"last_error": C.c_char_p, "workspace_bytes": _i64, "attn_workspace_bytes": _i64}
_OPTIONAL = {}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


STATUS_MAGNITUDE = 0x40    # PH_STATUS_MAGNITUDE
ROUTE_WIN_ALWAYS, ROUTE_WIN_NEVER, ROUTE_WIDE_ALWAYS, ROUTE_WIDE_NEVER, ROUTE_LIN_NEVER, ROUTE_GRID_NEVER = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20   # PH_ROUTE_*


class CBackend:
    """Thin typed wrapper over one shared library exporting the pasco_hip.h ABI."""

    def __init__(self, path: str, prefix: str, device_type: str):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found - build it first (python -m pasco_amd.build); "
                "pasco_amd has no CPU fallback")
        self.path = path
        self.prefix = prefix
        self.device_type = device_type
        self._serves_cuda = device_type == "cuda"
        self.lib = C.CDLL(path)
        self.fn: Dict[str, object] = {}
        # handshake BEFORE anything else is bound: version, then the size of the one struct that crosses the boundary
        ver = getattr(self.lib, prefix + "abi_version", None)
        v = int(ver()) if ver is not None else -1
        if v != ABI_VERSION:
            raise RuntimeError(f"{path}: C-ABI version {v}, this binding expects {ABI_VERSION} - rebuild the library "
                               "(python -m pasco_amd.build --force; oracle: python -m oracle.build)")
        dsz = getattr(self.lib, prefix + "conv_desc_size", None)
        if dsz is None or int(dsz()) != C.sizeof(ConvDesc):
            raise RuntimeError(f"{path}: sizeof(ph_conv_desc) = {None if dsz is None else int(dsz())}, the binding's mirror has "
                               f"{C.sizeof(ConvDesc)} bytes - rebuild the library against include/pasco_hip.h")
        for name, argtypes in _SIGNATURES.items():
            f = getattr(self.lib, prefix + name)
            f.argtypes = argtypes
            f.restype = _RESTYPES.get(name, C.c_int)
            self.fn[name] = f
        for name, argtypes in _OPTIONAL.items():
            f = getattr(self.lib, prefix + name, None)
            if f is not None:
                f.argtypes = argtypes
                f.restype = C.c_int
                self.fn[name] = f
        self._ws: Dict[torch.device, torch.Tensor] = {}
        self._status_ptrs: Dict[tuple, int] = {}         # (device type, index, stream handle) -> address of the status pair
        self._tls = threading.local()
        # tests may let a CPU checker library take the split-operand (mode 2) descriptors as well, so that the
        # host-side plumbing of that path (operand emission, split-only tensors) is exercised without a GPU
        self.checker_split = False

    # -- helpers -----------------------------------------------------------------------------------
    def routing(self, bits: int):
        """Context manager for parity tests: every `conv_fwd` of THIS thread inside it carries `ph_conv_desc.route = bits`
        (ROUTE_* below = PH_ROUTE_* of include/pasco_hip.h: pin the window / gather, wide / not, row-stream / not kernel of a shape
        that several kernels can serve).  Per call, no process-global state; the product path never sets it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = getattr(self._tls, "route", 0)
            self._tls.route = int(bits)
            try:
                yield self
            finally:
                self._tls.route = old
        return cm()

    def set_route(self, bits: int) -> int:
        """`ph_conv_desc.route` of this thread's following `conv_fwd` calls (see `routing`); returns the previous value."""
        old = getattr(self._tls, "route", 0)
        self._tls.route = int(bits)
        return old

    def has(self, name: str) -> bool:
        return name in self.fn

    def _check(self, rc: int, name: str):
        if rc != 0:
            msg = self.fn["last_error"]()
            raise RuntimeError(f"{self.prefix}{name} failed ({rc}): {msg.decode() if msg else ''}")

    def stream(self, device: torch.device) -> Optional[int]:
        """Raw handle of torch's current stream on `device` (every launch goes there).  ~430 calls per step: the raw getter
        (0.3 us) instead of building a torch.cuda.Stream object (4 us)."""
        if device.type == "cuda":
            idx = device.index
            if idx is None:
                idx = torch.cuda.current_device()
            return _raw_stream(idx)
        return None

    def _stream_key(self, device: torch.device):
        """Scratch buffers are reused from launch to launch, which is only safe in stream order: one set per
        (device, stream)."""
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return (device, self.stream(device) or 0)

    def release_stream(self, stream) -> int:
        """Drop every per-stream buffer (workspaces, split-K scratch, attention partials, the status pair) this backend keeps
        for `stream` (a torch.cuda.Stream or a raw handle): call it when a serving loop retires a stream - the buffers are
        keyed by the raw handle, so they would otherwise outlive it, and a later stream that is handed the same handle would
        inherit them together with any status bit nobody read.  Returns the number of buffers dropped."""
        handle = int(getattr(stream, "cuda_stream", stream) or 0)
        dev = getattr(stream, "device", None)
        if dev is None and torch.cuda.is_available():      # a raw handle: the streams of the current device
            dev = torch.device("cuda", torch.cuda.current_device())
        # per-stream entries are exactly the (name, device, handle) keys `_stream_key` makes: nothing else is touched
        drop = [k for k in self._ws
                if isinstance(k, tuple) and len(k) == 3 and isinstance(k[0], str) and isinstance(k[1], torch.device)
                and k[2] == handle and (dev is None or k[1] == dev)]
        for k in drop:
            del self._ws[k]
        for k in [k for k in self._status_ptrs if k[2] == handle and (dev is None or k[1] == dev.index)]:
            del self._status_ptrs[k]
        return len(drop)

    def workspace(self, n: int, device: torch.device) -> torch.Tensor:
        need = int(self.fn["workspace_bytes"](int(n)))
        key = ("ws",) + self._stream_key(device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def _chk(self, t: torch.Tensor, dtype, name: str):
        # ~900 calls per step: the passing case is three attribute tests, no torch.device object
        if t.dtype is dtype and t.is_cuda == self._serves_cuda and t.is_contiguous():
            return
        if t.dtype != dtype:
            raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        if t.device.type != self.device_type:
            raise ValueError(f"{name}: tensor on {t.device}, backend serves {self.device_type}")

    # -- coordinate maps ---------------------------------------------------------------------------
    @staticmethod
    def table_capacity(n: int) -> int:
        cap = 16
        while cap < 2 * n:
            cap <<= 1
        return cap

    def map_insert(self, coords: torch.Tensor, dedup: bool = True):
        """coords int32 [N,4] -> (tkeys, tvals, row2uniq|None, uniq_rows|None, n_uniq)."""
        tkeys, tvals, row2uniq, uniq_rows, n_uniq = self.map_insert_launch(coords, dedup)
        if not dedup:
            return tkeys, tvals, None, None, coords.shape[0]
        nu = int(n_uniq.item())
        return tkeys, tvals, row2uniq, uniq_rows[:nu], nu

    def map_insert_launch(self, coords: torch.Tensor, dedup: bool = True, n_uniq: Optional[torch.Tensor] = None):
        """The launches of `map_insert` without its host read: -> (tkeys, tvals, row2uniq [N], uniq_rows [N] (the first
        n_uniq entries valid), n_uniq int32 device tensor [1]); several independent inserts can then share ONE read of their
        counts (`n_uniq` = a one-element slice of a caller's count vector).  Each call takes its own scratch."""
        self._chk(coords, torch.int32, "coords")
        n = coords.shape[0]
        dev = coords.device
        cap = self.table_capacity(n)
        # keys and values in ONE allocation, values right behind the keys: the library then clears both with one fill
        table = torch.empty(cap + cap // 2, dtype=torch.int64, device=dev)
        tkeys = table[:cap]
        tvals = table[cap:].view(torch.int32)
        ws = torch.empty(int(self.fn["workspace_bytes"](int(n))), dtype=torch.uint8, device=dev)
        if dedup:
            # row2uniq | uniq_rows | the count: one allocation; the library WRITES the count (no zero fill needed)
            m = max(n, 1)
            m4 = (m + 3) // 4 * 4                     # every part on a 16-byte boundary
            rows2 = torch.empty(2 * m4 + 1, dtype=torch.int32, device=dev)
            row2uniq, uniq_rows = rows2[:m], rows2[m4:m4 + m]
            if n_uniq is None:
                n_uniq = rows2[2 * m4:]
            rc = self.fn["map_insert"](_ptr(coords), n, _ptr(tkeys), _ptr(tvals), cap, _ptr(row2uniq),
                                       _ptr(uniq_rows), n_uniq.data_ptr(), _ptr(ws), ws.numel(), self.status_ptr(dev),
                                       self.stream(dev))
            self._check(rc, "map_insert")
            return tkeys, tvals, row2uniq[:n], uniq_rows, n_uniq
        rc = self.fn["map_insert"](_ptr(coords), n, _ptr(tkeys), _ptr(tvals), cap, None, None, None,
                                   _ptr(ws), ws.numel(), self.status_ptr(dev), self.stream(dev))
        self._check(rc, "map_insert")
        return tkeys, tvals, None, None, None

    def map_find(self, query: torch.Tensor, tkeys: torch.Tensor, tvals: torch.Tensor) -> torch.Tensor:
        self._chk(query, torch.int32, "query")
        n = query.shape[0]
        out = torch.empty(n, dtype=torch.int32, device=query.device)
        rc = self.fn["map_find"](_ptr(query), n, _ptr(tkeys), _ptr(tvals), tkeys.numel(), _ptr(out),
                                 self.stream(query.device))
        self._check(rc, "map_find")
        return out

    def coords_floor(self, coords: torch.Tensor, ts: int) -> torch.Tensor:
        self._chk(coords, torch.int32, "coords")
        out = torch.empty_like(coords)
        rc = self.fn["coords_floor"](_ptr(coords), coords.shape[0], int(ts), _ptr(out),
                                     self.stream(coords.device))
        self._check(rc, "coords_floor")
        return out

    def coords_expand(self, coords: torch.Tensor, ts_out: int) -> torch.Tensor:
        self._chk(coords, torch.int32, "coords")
        n = coords.shape[0]
        out = torch.empty((n * 8, 4), dtype=torch.int32, device=coords.device)
        rc = self.fn["coords_expand"](_ptr(coords), n, int(ts_out), _ptr(out), self.stream(coords.device))
        self._check(rc, "coords_expand")
        return out

    def nbr_build(self, out_coords: torch.Tensor, tkeys: torch.Tensor, tvals: torch.Tensor,
                  offsets, same_map: bool = False) -> torch.Tensor:
        """offsets: list of (dx,dy,dz) already scaled by the tensor stride.  `same_map`: out_coords are the (unique) rows
        the table was built from and the offsets are symmetric -> half the probes (ph_nbr_build_same)."""
        self._chk(out_coords, torch.int32, "out_coords")
        kvol = len(offsets)
        if not 1 <= kvol <= MAX_KVOL:
            raise ValueError(f"kernel volume {kvol} not supported (max {MAX_KVOL})")
        n_out = out_coords.shape[0]
        flat = (_i32 * (3 * kvol))(*[int(v) for o in offsets for v in o])
        nbr = torch.empty((kvol, n_out), dtype=torch.int32, device=out_coords.device)
        name = "nbr_build_same" if same_map else "nbr_build"
        rc = self.fn[name](_ptr(out_coords), n_out, _ptr(tkeys), _ptr(tvals), tkeys.numel(),
                           C.cast(flat, _vp), kvol, _ptr(nbr), self.stream(out_coords.device))
        self._check(rc, name)
        return nbr

    def rowlist_build(self, nbr: torch.Tensor) -> dict:
        """Padded row lists of a kernel map whose output rows have exactly one pair each (generative transposed
        convolutions): {"in", "out"} int32 [cap], {"tile_k"} int32 [cap / 128] (include/pasco_hip.h ph_conv_desc.rl_*).  No
        host read: sizes are upper bounds, counts stay on the device."""
        self._chk(nbr, torch.int32, "nbr")
        kvol, n_out = nbr.shape
        pin, pout, counts = self.kmap_compact(nbr)
        # the caller's promise "exactly one pair per output row": a map that breaks it would leave output rows unwritten.
        # Checked on the device by the pack kernel itself (pairs != rows -> status bit 5, reported by the end-of-step check)
        cap = (n_out + kvol * 127 + 127) // 128 * 128
        tcap = cap // 128
        dev = nbr.device
        rl_in = torch.empty(cap, dtype=torch.int32, device=dev)
        rl_out = torch.empty(cap, dtype=torch.int32, device=dev)
        tile_k = torch.empty(tcap, dtype=torch.int32, device=dev)
        rc = self.fn["rowlist_pack"](_ptr(pin), _ptr(pout), _ptr(counts), kvol, n_out, _ptr(rl_in), _ptr(rl_out), _ptr(tile_k),
                                     cap, tcap, self.status_ptr(dev), self.stream(dev))
        self._check(rc, "rowlist_pack")
        return {"in": rl_in, "out": rl_out, "tile_k": tile_k}

    def win_build(self, nbr: torch.Tensor) -> dict:
        """Input windows of a 3x3x3 kernel map (include/pasco_hip.h ph_win_build): per tile of 128 consecutive output rows
        the ascending list of distinct input rows (`rows` [T, 3456], `cnt` [T]) and the position of every (offset, row)
        entry in it (`slots` uint16 [T, 27, 128], 0xFFFF = none); `stats` = device-side pass counts.  Cached by the
        coordinate manager next to `nbr`."""
        self._chk(nbr, torch.int32, "nbr")
        kvol, n_out = nbr.shape
        if kvol != 27:
            raise ValueError("win_build serves 3x3x3 kernel maps")
        dev = nbr.device
        t = (n_out + 127) // 128
        win = {"rows": torch.empty((max(t, 1), 27 * 128), dtype=torch.int32, device=dev),
               # cnt: every tile is written by the build; stats: cleared by ph_win_build itself (n_out == 0: nothing to clear)
               "cnt": torch.empty(max(t, 1), dtype=torch.int32, device=dev) if n_out > 0 else torch.zeros(1, dtype=torch.int32, device=dev),
               "slots": torch.empty((max(t, 1), 27, 128), dtype=torch.int16, device=dev),     # uint16 bit patterns
               "stats": torch.empty(4, dtype=torch.int32, device=dev) if n_out > 0 else torch.zeros(4, dtype=torch.int32, device=dev),
               "n_out": n_out}
        rc = self.fn["win_build"](_ptr(nbr), kvol, n_out, _ptr(win["rows"]), _ptr(win["cnt"]), _ptr(win["slots"]),
                                  _ptr(win["stats"]), self.stream(dev))
        self._check(rc, "win_build")
        return win

    def kmap_compact(self, nbr: torch.Tensor):
        """-> (pairs_in [K,N], pairs_out [K,N], counts [K]) ; segment k valid up to counts[k]."""
        self._chk(nbr, torch.int32, "nbr")
        kvol, n_out = nbr.shape
        dev = nbr.device
        pin = torch.empty_like(nbr)
        pout = torch.empty_like(nbr)
        counts = torch.empty(kvol, dtype=torch.int32, device=dev)       # written by the library's scan
        ws = self.workspace(n_out, dev)
        rc = self.fn["kmap_compact"](_ptr(nbr), kvol, n_out, _ptr(pin), _ptr(pout), _ptr(counts),
                                     _ptr(ws), ws.numel(), self.stream(dev))
        self._check(rc, "kmap_compact")
        return pin, pout, counts

    # -- convolution -------------------------------------------------------------------------------
    def conv_fwd(self, x: Optional[torch.Tensor], weight: Optional[torch.Tensor], nbr: Optional[torch.Tensor],
                 n_out: int, *, wshape=None, xshape=None, bias=None, pro_scale=None, pro_shift=None, pro_act=ACT_NONE, epi_scale=None,
                 epi_shift=None, epi_act=ACT_NONE, slope=0.01, residual=None, res_act=ACT_NONE,
                 epi2_scale=None, epi2_shift=None, split=None, in_split: Optional[torch.Tensor] = None,
                 emit_split=None, want_out: bool = True,
                 out: Optional[torch.Tensor] = None, win=None, in_split_has_prologue: bool = False, axis=None,
                 rowlist=None, out_split: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
                 exact_if: Optional[torch.Tensor] = None, grid=None):
        """out = epilogue(sum_k gather(prologue(x))[k] @ W[k]) - one `ph_conv_fwd` launch (include/pasco_hip.h).

        `split` selects the split-precision products: (w_hi, w_lo, unscale) from `split_weight_f16` = mode 1
        (activations split inside the kernel), (w_split, unscale) from `split_weight_rows` = mode 2 (both operands
        pre-split; `in_split` = `split_rows(x, prologue)` or computed here).  In mode 2 `x` / `weight` may be None
        (`xshape` = (n_in, cin) / `wshape` = (kvol, cin, cout)) when only the operands exist.
        `emit_split` = (scale | None, shift | None, act) (mode 2, cout % 32 == 0): also write the split operand
        of act(out * scale + shift) for the next convolution and return (out, out_split); `want_out=False` then
        skips the fp32 result (returns (None, out_split)).
        `win` = `win_build(nbr)` (3x3x3 maps, mode 2): lets the library serve the launch from LDS-resident input windows
        where the map is local enough (decided on the device).
        `axis` = (table fp32 [3, T, cout], coords int32 [n_out, 4], lo) (mode 2): the per-axis table residual
        table[0][x - lo] + table[1][y - lo] + table[2][z - lo] is added where `residual` is added.
        `grid` = ((B, X, Y, Z), (kx, ky, kz)) (mode 2): the PROMISE that `nbr` is the stride-1 box kernel map of the full dense
        grid with sites (b, z, x, y) and offsets y-fastest (`grid_offsets`; ph_conv_desc.grid_dims): served from LDS windows.
        `status` (int32 [1] device word): the launch reports its flags there instead of into the stream's status pair;
        `exact_if` (exact fp32 launches only): the launch does its work only when bit 0 of that word is set - the guarded
        form of the split path (include/pasco_hip.h ph_conv_desc.exact_if; `pasco_amd.me.modules`)."""
        if x is None:          # rows that exist only as a pre-split operand (mode 2): `xshape` = (n_in, cin)
            if xshape is None or in_split is None or split is None or len(split) != 2 or not self.split_capable():
                raise ValueError("conv: x=None needs xshape, in_split and a mode-2 split on the device backend")
            if pro_scale is not None or pro_shift is not None or pro_act != ACT_NONE:
                raise ValueError("conv: a pre-split input already carries its prologue")
            x = torch.empty((xshape[0], xshape[1]), dtype=torch.float32, device="meta")   # shape carrier only
            dev = in_split.device
        else:
            self._chk(x, torch.float32, "in")
            dev = x.device
        if weight is None:     # pre-split operands only (mode 2): the fp32 kernel is not read, `wshape` = (kvol, cin, cout)
            if wshape is None or split is None or len(split) != 2 or not self.split_capable():
                raise ValueError("conv: weight=None needs wshape and a mode-2 split on the device backend")
            kvol, cin, cout = wshape
        else:
            self._chk(weight, torch.float32, "weight")
            if weight.dim() == 2:
                kvol, (cin, cout) = 1, weight.shape
            else:
                kvol, cin, cout = weight.shape
        if x.shape[1] != cin:
            raise ValueError(f"conv: input has {x.shape[1]} channels, kernel expects {cin}")
        if nbr is not None:
            self._chk(nbr, torch.int32, "nbr")
            if tuple(nbr.shape) != (kvol, n_out):
                raise ValueError(f"conv: nbr shape {tuple(nbr.shape)} != {(kvol, n_out)}")
        emit = emit_split is not None and split is not None and len(split) == 2 and cout % 32 == 0
        if emit_split is not None and not emit:
            raise ValueError("conv: emit_split needs a mode-2 split and cout % 32 == 0")
        if out is None and (want_out or not emit):
            out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
        if emit and out_split is not None:       # caller-owned operand rows (a row slice of a larger operand tensor)
            if out_split.dtype != torch.float16 or not out_split.is_contiguous() or out_split.numel() != n_out * 2 * cout:
                raise ValueError("conv: out_split must be contiguous f16 [n_out, cout / 32, 2, 32]")
        else:
            out_split = torch.empty((n_out, cout // 32, 2, 32), dtype=torch.float16, device=dev) if emit else None
        if n_out == 0:
            return (out, out_split) if emit else out
        d = ConvDesc()
        d.route = getattr(self._tls, "route", 0)
        if emit:
            osc, osh, oact = emit_split
            for name, t in (("osp_scale", osc), ("osp_shift", osh)):
                if t is not None:
                    self._chk(t, torch.float32, name)
                    if t.numel() != cout:
                        raise ValueError(f"conv: {name} has {t.numel()} entries, expected {cout}")
            d.out_split, d.osp_scale, d.osp_shift, d.osp_act = _ptr(out_split), _ptr(osc), _ptr(osh), int(oact)
        d.in_, d.weight, d.nbr, d.out = (None if x.is_meta else _ptr(x)), _ptr(weight), _ptr(nbr), _ptr(out)
        d.n_in, d.n_out, d.cin, d.cout, d.kvol = x.shape[0], n_out, cin, cout, kvol
        d.pro_act, d.epi_act, d.res_act, d.epi_slope = pro_act, epi_act, res_act, float(slope)
        for name, t, c in (("pro_scale", pro_scale, cin), ("pro_shift", pro_shift, cin),
                           ("bias", bias, cout), ("epi_scale", epi_scale, cout),
                           ("epi_shift", epi_shift, cout), ("epi2_scale", epi2_scale, cout),
                           ("epi2_shift", epi2_shift, cout)):
            if t is not None:
                self._chk(t, torch.float32, name)
                if t.numel() != c:
                    raise ValueError(f"conv: {name} has {t.numel()} entries, expected {c}")
            setattr(d, name, _ptr(t))
        if residual is not None:
            self._chk(residual, torch.float32, "residual")
            if tuple(residual.shape) != (n_out, cout):
                raise ValueError("conv: residual shape mismatch")
        d.residual = _ptr(residual)
        if axis is not None:
            tab, acoords, lo = axis
            self._chk(tab, torch.float32, "axis table")
            self._chk(acoords, torch.int32, "axis coords")
            if tab.dim() != 3 or tab.shape[0] != 3 or tab.shape[2] != cout or tuple(acoords.shape) != (n_out, 4):
                raise ValueError("conv: axis = (table [3, T, cout], coords [n_out, 4], lo)")
            if self.device_type == "cuda" and (split is None or len(split) != 2):
                raise ValueError("conv: the axis-table residual is served by the pre-split (mode 2) path")
            d.axis_table, d.axis_coords, d.axis_lo, d.axis_rows = _ptr(tab), _ptr(acoords), int(lo), int(tab.shape[1])
        if split is not None:      # opt-in f16x3 products
            if len(split) == 2:    # (w_split, unscale) from split_weight_rows + in_split from split_rows: mode 2
                w_split, unscale = split
                if in_split is None:
                    in_split = self.split_rows(x, pro_scale=pro_scale, pro_shift=pro_shift, pro_act=pro_act, slope=slope)
                elif not x.is_meta and (pro_scale is not None or pro_shift is not None or pro_act != ACT_NONE) and \
                        not in_split_has_prologue:
                    # the device library never applies a prologue in mode 2: a caller-made operand must already carry it
                    raise ValueError("conv: in_split given together with a prologue - pass in_split_has_prologue=True if "
                                     "the operand was built with split_rows(x, pro_scale=..., pro_shift=..., pro_act=...), "
                                     "else drop in_split")
                cpad = (cin + 31) // 32 * 32
                if in_split.dtype != torch.float16 or in_split.numel() != x.shape[0] * 2 * cpad:
                    raise ValueError("conv: in_split does not match the input rows")
                if w_split.numel() != kvol * cout * 2 * cpad:
                    raise ValueError("conv: w_split does not match the kernel")
                d.mma_mode, d.in_split, d.w_split = 2, _ptr(in_split), _ptr(w_split)
                if grid is not None and nbr is not None:
                    gd, gk = grid
                    if len(gd) != 4 or len(gk) != 3 or gd[0] * gd[1] * gd[2] * gd[3] != n_out or x.shape[0] != n_out or \
                            gk[0] * gk[1] * gk[2] != kvol or any(k % 2 == 0 for k in gk):
                        raise ValueError("conv: grid = ((B, X, Y, Z), (kx, ky, kz)) must describe the whole map (odd kernel sizes)")
                    for q in range(4):
                        d.grid_dims[q] = int(gd[q])
                    for q in range(3):
                        d.grid_kernel[q] = int(gk[q])
                if rowlist is not None and nbr is not None:     # one-pair-per-row map: k = 1 products per list tile
                    d.rl_in, d.rl_out, d.rl_tile_k = _ptr(rowlist["in"]), _ptr(rowlist["out"]), _ptr(rowlist["tile_k"])
                    d.rl_rows, d.rl_tiles = int(rowlist["in"].numel()), int(rowlist["tile_k"].numel())
                if win is not None and kvol == 27 and nbr is not None:
                    d.win_rows, d.win_cnt, d.win_slots, d.win_stats = (_ptr(win["rows"]), _ptr(win["cnt"]),
                                                                       _ptr(win["slots"]), _ptr(win["stats"]))
                    if 32 < cout <= 64 and self._serves_cuda:       # the window kernel of the 64-wide outputs reads its weights in
                        d.w_frag = _ptr(self.weight_fragments(w_split, kvol, cout, cpad))     # fragment order (cached on w_split)
            else:                  # (w_hi, w_lo, unscale) from split_weight_f16: mode 1, activations split in-kernel
                w_hi, w_lo, unscale = split
                d.mma_mode, d.w_f16_hi, d.w_f16_lo = 1, _ptr(w_hi), _ptr(w_lo)
            d.split_exp2 = SPLIT_ACT_EXP2
            d.w_unscale = float(unscale) * 2.0 ** (-SPLIT_ACT_EXP2)
            d.status = self.status_ptr(dev) if status is None else status.data_ptr()
            if kvol > 1 or n_out * cout <= (1 << 23):
                # scratch for a split over the kernel offsets: few-row layers split whole (up to 12 partial copies), big maps only
                # their last partial round of row tiles (ph_conv_dma_try's tail split: slices x tail tiles <= the 512 resident
                # 128 x 128 tiles of one round = 32 MiB of fp32 partial sums whatever the width)
                need = 12 * n_out * cout * 4 if n_out * cout <= (1 << 23) else 512 * 128 * 128 * 4
                key = ("splitk",) + self._stream_key(dev)
                sk = self._ws.get(key)
                if sk is None or sk.numel() < need:
                    sk = torch.empty(need, dtype=torch.uint8, device=dev)
                    self._ws[key] = sk
                d.splitk_ws, d.splitk_ws_bytes = _ptr(sk), sk.numel()
        if exact_if is not None:
            if split is not None:
                raise ValueError("conv: exact_if guards an exact fp32 launch (no split)")
            d.exact_if = exact_if.data_ptr()
        rc = self.fn["conv_fwd"](C.byref(d), self.stream(dev))
        self._check(rc, "conv_fwd")
        return (out, out_split) if emit else out

    @staticmethod
    def grid_offsets(kernel_size) -> list:
        """Offsets of a (kx, ky, kz) box in the enumeration the dense-grid promise of `conv_fwd(grid=...)` names: k = iy + ky * (ix + kx * iz)
        - y fastest, like the sites (b, z, x, y) of the grid (include/pasco_hip.h ph_conv_desc.grid_kernel)."""
        kx, ky, kz = (int(k) for k in kernel_size)
        return [(x, y, z) for z in range(-(kz // 2), kz // 2 + 1) for x in range(-(kx // 2), kx // 2 + 1)
                for y in range(-(ky // 2), ky // 2 + 1)]

    def conv_last_config(self) -> dict:
        """Which kernel instantiation the last `conv_fwd` of this thread launched (include/pasco_hip.h
        ph_conv_last_config): the parity tests assert that the instantiations the benchmark runs are the ones they
        compared with the oracle."""
        buf = (_i32 * 8)()
        self._check(self.fn["conv_last_config"](C.cast(buf, _vp)), "conv_last_config")
        names = ("mma_mode", "bm", "bn", "kc", "ksplit", "emit", "kernel", "waves")
        return dict(zip(names, [int(v) for v in buf]))

    def status_word(self, device) -> torch.Tensor:
        """Device word the kernels OR their flags into: one per (device, STREAM), like the scratch buffers - every kernel
        that can raise a flag of a scene and the read-and-clear of `check_status` are then ordered by the stream, so scenes
        in flight on other streams neither lose a flag nor see a foreign one."""
        return self._status_pair(device)[0:1]

    def status_ptr(self, device) -> int:
        """Address of `status_word(device)` - what the launches take.  ~260 calls per step: the address of a stream's pair is
        looked up by (device index, raw stream handle) without building a torch.device, a key tuple or a one-element view."""
        pinned = getattr(self._tls, "status_pin", None)
        if pinned is not None:
            return pinned.data_ptr()
        idx = device.index
        if idx is None and device.type == "cuda":
            idx = torch.cuda.current_device()
        k = (device.type, idx, _raw_stream(idx) if device.type == "cuda" else 0)
        hit = self._status_ptrs.get(k)
        if hit is None:
            hit = self._status_pair(device).data_ptr()
            self._status_ptrs[k] = hit
        return hit

    def optimistic_word(self, device) -> torch.Tensor:
        """Second word of the stream's status pair: kernels and torch ops OR a non-zero value into it when an OPTIMISTIC
        shortcut of the graph turned out not to hold (a fast path taken without the host read that would have justified it
        - see `pasco_amd.graph.fused.optimistic`).  `check_status` reports it as bit 4 (16) and the caller redoes the step
        with the shortcuts off."""
        return self._status_pair(device)[1:2]

    def _status_pair(self, device) -> torch.Tensor:
        pinned = getattr(self._tls, "status_pin", None)
        if pinned is not None:
            return pinned
        key = ("status",) + self._stream_key(torch.device(device))
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(2, dtype=torch.int32, device=device)
            self._ws[key] = t
        return t

    @contextlib.contextmanager
    def pin_status(self, device):
        """Inside the block every launch of THIS thread reports into the status word of the stream that is current NOW -
        for work that is recorded on another stream but belongs to this one (hipGraph capture runs on a capture stream;
        the replay is launched on, and checked from, the caller's stream)."""
        prev = getattr(self._tls, "status_pin", None)
        self._tls.status_pin = None
        self._tls.status_pin = self._status_pair(device)
        try:
            yield
        finally:
            self._tls.status_pin = prev

    STATUS_TEXT = {
        1: "an activation outside the f16 range in a split-precision convolution / attention operand (rerun with "
           "pasco_amd.graph.fused.set_conv_precision('f32'); PascoNet.forward does so by itself)",
        2: "a coordinate outside the packable range (batch index 0..1023, coordinates -131072..131071) was inserted into a "
           "coordinate map; it would alias another voxel",
        32: "a kernel map handed over as one-pair-per-output-row (fused.conv(..., one_pair=True): row lists of a generative "
            "transposed convolution) has another number of pairs than rows: output rows would stay unwritten (PASCO_CONV_RL=0)",
        16: "an optimistic shortcut of the graph did not hold (a fast path taken without the host read that would justify it: "
            "attention-mask block lookups with a coordinate outside its subnet's box, kept rows that are not the leading rows, "
            "an all-zero bottleneck site); PASCO_OPTIMISTIC=0 takes the checked paths; PascoNet.forward redoes the step by itself",
        8: "the fused input stage met a merged row that is entirely zero or a point outside its box (ME.to_sparse drops such a "
           "row: the stage has to run on its general path, PASCO_INPUT_FUSED=0; PascoNet.forward redoes it by itself)",
        4: "a coordinate outside the rows of a per-axis table residual (ph_conv_desc.axis_table) was clamped to the table's "
           "edge; the materialised forms are selected with PASCO_RESIZE_ABSORB=0 PASCO_PE_TABLE=0 PASCO_HEAD_ABSORB=0",
    }

    def check_status(self, device) -> None:
        """Raise if, since the last check ON THIS STREAM, a kernel raised a status bit: 0 = a value outside the f16 range
        met a split-precision operand, 1 = a coordinate map was given a coordinate its 64-bit key cannot hold, 2 = a
        per-axis table clamped a coordinate.  One device->host read (call where the host synchronises anyway); read and
        clear are two stream-ordered operations on the stream's own word.  Every raised bit is reported; the exception is
        an `F16RangeError` when bit 0 is the only one (the caller may redo the step on the exact path)."""
        t = self._ws.get(("status",) + self._stream_key(torch.device(device)))
        if t is None:
            return
        snap = t.clone()          # stream-ordered: after every kernel of this stream that could raise a flag ...
        t.zero_()                 # ... and before any later one
        v, opt = snap.tolist()
        v &= ~STATUS_MAGNITUDE        # informational (ph_split_rows: the operand holds a full-precision value): not an error
        if opt != 0:
            v |= 16
        if v == 0:
            return
        msgs = [self.STATUS_TEXT[b] for b in (1, 2, 4, 8, 16, 32) if v & b]
        if v & ~63:
            msgs.append(f"unknown status bits {v & ~63:#x}")
        text = f"pasco_amd: device status {v:#x}: " + "; ".join(msgs)
        raise (F16RangeError if v == 1 else StatusError)(v, text)

    @staticmethod
    def split_weight_f16(weight: torch.Tensor, exponent=None):
        """fp32 kernel [K, cin, cout] (or [cin, cout]) -> (hi, lo) f16 [K, cout, cin] of weight * 2^e and the
        factor 2^-e.  e puts the largest magnitude just below 2^14 so that every lo part is a normal f16
        (one device->host read; static weights are split once and cached).  Pass `exponent` to skip the
        read for per-call operands of known magnitude."""
        w = weight.detach().float()
        if w.dim() == 2:
            w = w[None]
        if exponent is not None:
            e = int(exponent)
        else:
            wmax = float(w.abs().max())
            e = 0 if wmax == 0.0 else 13 - int(torch.frexp(torch.tensor(wmax))[1])
        scaled = (w * float(2.0 ** e)).transpose(1, 2).contiguous()   # exact power of two (torch.ldexp goes through pow on the GPU: inexact)
        hi = scaled.to(torch.float16)
        lo = (scaled - hi.float()).to(torch.float16)
        return hi.contiguous(), lo.contiguous(), float(2.0 ** (-e))

    def split_rows(self, x: torch.Tensor, *, pro_scale=None, pro_shift=None, pro_act=ACT_NONE, slope=0.01,
                   out: Optional[torch.Tensor] = None, exp2: Optional[int] = None,
                   status: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 rows [n, c] -> f16 [n, cpad/32, 2, 32] (hi | lo groups) of act(x * scale + shift) * 2^exp2: the
        operand layout of mma_mode 2 (include/pasco_hip.h ph_split_rows).  exp2 defaults to SPLIT_ACT_EXP2 (an
        activation operand, what `conv_fwd` expects as `in_split`); pre-scaled weights pass 0."""
        exp2 = SPLIT_ACT_EXP2 if exp2 is None else int(exp2)
        self._chk(x, torch.float32, "in")
        n, c = x.shape
        if c % 8 != 0:
            raise ValueError("split_rows: needs c % 8 == 0")
        cpad = (c + 31) // 32 * 32
        if out is None:
            out = torch.empty((n, cpad // 32, 2, 32), dtype=torch.float16, device=x.device)
        for name, t in (("pro_scale", pro_scale), ("pro_shift", pro_shift)):
            if t is not None:
                self._chk(t, torch.float32, name)
                if t.numel() != c:
                    raise ValueError(f"split_rows: {name} has {t.numel()} entries, expected {c}")
        rc = self.fn["split_rows"](_ptr(x), n, c, _ptr(pro_scale), _ptr(pro_shift), pro_act, float(slope), exp2, _ptr(out),
                                   self.status_ptr(x.device) if status is None else status.data_ptr(), self.stream(x.device))
        self._check(rc, "split_rows")
        return out

    def split_weight_rows(self, weight: torch.Tensor, exponent=None):
        """fp32 kernel [K, cin, cout] (or [cin, cout]) -> (w_split, 2^-e): the mode-2 operand of weight * 2^e,
        transposed to [K, cout, cin] rows and split by `split_rows` (static weights: once, cached by callers)."""
        w = weight.detach().float()
        if w.dim() == 2:
            w = w[None]
        if exponent is not None:
            e = int(exponent)
        else:
            wmax = float(w.abs().max())
            e = 0 if wmax == 0.0 else 13 - int(torch.frexp(torch.tensor(wmax))[1])
        k, cin, cout = w.shape
        rows = (w * float(2.0 ** e)).transpose(1, 2).contiguous().view(k * cout, cin)   # exact power of two
        return self.split_rows(rows, exp2=0), float(2.0 ** (-e))

    @staticmethod
    def weight_fragments(w_split: torch.Tensor, kvol: int, cout: int, cpad: int) -> torch.Tensor:
        """`ph_conv_desc.w_frag`: the rows of `w_split` (f16 [kvol * cout, cpad / 32, 2, 32]) in the fragment order of the 64-wide
        window kernel - f16 [kvol, cpad / 16, 2 (column block), 2 (hi, lo), 64 (lane = l31 + 32 h), 8]: lane's 8 channels
        16 c + 8 h .. + 7 of column min(32 j + l31, cout - 1).  A pure permutation, made once per kernel tensor (kept on it)."""
        hit = getattr(w_split, "_ph_wfrag", None)
        if hit is not None and hit[0] == (w_split._version, kvol, cout, cpad):
            return hit[1]
        g = cpad // 32
        ws = w_split.view(kvol, cout, g, 2, 2, 2, 8)               # [k, n, group, part, s = chunk parity, h, q]
        rows = torch.arange(64, device=w_split.device).clamp_(max=cout - 1)
        ws = ws.index_select(1, rows).view(kvol, 2, 32, g, 2, 2, 2, 8)     # [k, j, l31, group, part, s, h, q]
        frag = ws.permute(0, 3, 5, 1, 4, 6, 2, 7).contiguous()     # [k, group, s, j, part, h, l31, q]: chunk c = 2 group + s
        frag = frag.view(kvol, 2 * g, 2, 2, 64, 8)
        try:
            w_split._ph_wfrag = ((w_split._version, kvol, cout, cpad), frag)
        except AttributeError:
            pass
        return frag

    def split_capable(self) -> bool:
        return self.device_type == "cuda" or self.checker_split

    def split_supported(self, cin: int, cout: int) -> bool:
        return self.split_capable() and cin % 8 == 0 and cout % 4 == 0

    def maxpool_fwd(self, x: torch.Tensor, nbr: torch.Tensor) -> torch.Tensor:
        self._chk(x, torch.float32, "in")
        self._chk(nbr, torch.int32, "nbr")
        kvol, n_out = nbr.shape
        out = torch.empty((n_out, x.shape[1]), dtype=torch.float32, device=x.device)
        rc = self.fn["maxpool_fwd"](_ptr(x), x.shape[1], _ptr(nbr), kvol, n_out, _ptr(out),
                                    self.stream(x.device))
        self._check(rc, "maxpool_fwd")
        return out

    # -- rows --------------------------------------------------------------------------------------
    def mask_compact(self, mask: torch.Tensor) -> torch.Tensor:
        """bool/uint8 mask [N] -> int32 rows kept (order preserved)."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        self._chk(mask, torch.uint8, "mask")
        n = mask.shape[0]
        dev = mask.device
        buf = torch.empty(max(n, 1) + 1, dtype=torch.int32, device=dev)      # rows | count (written by the library)
        keep, cnt = buf[:-1], buf[-1:]
        ws = self.workspace(n, dev)
        rc = self.fn["mask_compact"](_ptr(mask), n, _ptr(keep), cnt.data_ptr(), _ptr(ws), ws.numel(),
                                     self.stream(dev))
        self._check(rc, "mask_compact")
        return keep[: int(cnt.item())]

    # -- input stage (include/pasco_hip.h points_* / cells_max) ---------------------------------------------------
    MAX_INPUT_SITES = 1 << 27      # byte flags + int32 ranks per site of the points' bounding box: 640 MB at the cap

    def pooled_merge(self, h: torch.Tensor, xyz: torch.Tensor, starts, bounds=None):
        """Voxel max of the point features + MIMO channel concatenation in one sort-free pass (CylinderFeat's scatter_max
        + Augmenter.merge): h fp32 [P, C] = the point MLP's output for the points of ALL subnets (subnet b's points are
        rows starts[b] .. starts[b + 1]), xyz int64 [P, 3] their voxel indices -> (coords int32 [V, 4] = (0, x, y, z) in
        lexicographic order, feats fp32 [V, M * C]).  Two host reads (bounding box - skipped with `bounds` = (lo3, hi3)
        host ints - and the row count).  An all-zero merged row (which ME.to_sparse would drop) raises status bit 3 for
        the caller's end-of-step check instead of costing a third read.  None when the box is too large for site flags."""
        self._chk(h, torch.float32, "h")
        self._chk(xyz, torch.int64, "xyz")
        n, c = h.shape
        m = len(starts) - 1
        if not (c % 4 == 0 and 1 <= m <= 8 and tuple(xyz.shape) == (n, 3)):
            raise ValueError("pooled_merge: h [P, C] with C % 4 == 0, xyz int64 [P, 3], at most 8 subnets")
        dev = h.device
        st = self.stream(dev)
        if bounds is None:
            b6 = torch.empty(6, dtype=torch.int32, device=dev)
            self._check(self.fn["points_bounds"](_ptr(xyz), n, _ptr(b6), st), "points_bounds")
            b6 = b6.tolist()
            lo, hi = b6[:3], b6[3:]
        else:
            lo, hi = [int(v) for v in bounds[0]], [int(v) for v in bounds[1]]
        dims = [hi[a] - lo[a] + 1 for a in range(3)]
        nsites = dims[0] * dims[1] * dims[2]
        if n == 0 or min(dims) <= 0 or nsites > self.MAX_INPUT_SITES:
            return None
        hlo, hdim = (_i32 * 3)(*lo), (_i32 * 3)(*dims)
        hst = (_i64 * (m + 1))(*[int(v) for v in starts])
        flags = torch.zeros(nsites, dtype=torch.uint8, device=dev)
        status = self.status_ptr(dev)
        self._check(self.fn["points_mark"](_ptr(xyz), n, C.cast(hlo, _vp), C.cast(hdim, _vp), _ptr(flags), status, st),
                    "points_mark")
        sites = torch.empty(min(n, nsites), dtype=torch.int32, device=dev)
        rank = torch.empty(nsites, dtype=torch.int32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        ws = self.workspace(nsites, dev)
        self._check(self.fn["mask_compact_rank"](_ptr(flags), nsites, _ptr(sites), _ptr(rank), _ptr(cnt), _ptr(ws), ws.numel(),
                                                 st), "mask_compact_rank")
        v = int(cnt.item())
        head = torch.full((max(v * m, 1),), -1, dtype=torch.int32, device=dev)
        nxt = torch.empty(n, dtype=torch.int32, device=dev)
        self._check(self.fn["points_link"](_ptr(xyz), n, C.cast(hst, _vp), m, C.cast(hlo, _vp), C.cast(hdim, _vp), _ptr(rank),
                                           _ptr(head), _ptr(nxt), st), "points_link")
        out = torch.empty((v, m * c), dtype=torch.float32, device=dev)
        coords = torch.empty((v, 4), dtype=torch.int32, device=dev)
        self._check(self.fn["cells_max"](_ptr(h), c, _ptr(head), _ptr(nxt), v, m, _ptr(sites), C.cast(hlo, _vp), C.cast(hdim, _vp),
                                         _ptr(out), _ptr(coords), status, st), "cells_max")
        return coords, out

    def mask_compact_many(self, masks) -> list:
        """`mask_compact` of several independent masks with ONE host read for all their counts (the compactions are
        launched back to back, each with its own scratch; the counts come back in one small copy) -> list of int32 row
        tensors.  What a sequence of prunes that do not depend on one another costs: one synchronisation, not one each."""
        masks = [m.view(torch.uint8) if m.dtype == torch.bool else m for m in masks]
        if not masks:
            return []
        dev = masks[0].device
        cnts = torch.empty(len(masks), dtype=torch.int32, device=dev)     # every entry written by its compaction
        keeps = []
        for j, mask in enumerate(masks):
            self._chk(mask, torch.uint8, "mask")
            n = mask.shape[0]
            keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            ws = torch.empty(int(self.fn["workspace_bytes"](int(n))), dtype=torch.uint8, device=dev)   # one per launch
            rc = self.fn["mask_compact"](_ptr(mask), n, _ptr(keep), cnts[j:j + 1].data_ptr(), _ptr(ws), ws.numel(),
                                         self.stream(dev))
            self._check(rc, "mask_compact")
            keeps.append(keep)
        return [k[:c] for k, c in zip(keeps, cnts.tolist())]

    def gather_rows(self, src: torch.Tensor, rows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """src [N,C] (4-byte dtype), rows int32 [M] -> [M,C]; rows == -1 give zeros.  `out`: caller-owned contiguous [M, C] rows
        (a slice of a batch buffer: no copy afterwards)."""
        if src.element_size() != 4:
            raise TypeError("gather_rows serves 4-byte element types")
        if not src.is_contiguous():
            raise ValueError("gather_rows: src must be contiguous")
        self._chk(rows, torch.int32, "rows")
        c = src.shape[1]
        if out is None:
            out = torch.empty((rows.shape[0], c), dtype=src.dtype, device=src.device)
        elif out.shape != (rows.shape[0], c) or out.dtype != src.dtype or not out.is_contiguous() or out.device != src.device:
            raise ValueError("gather_rows: out must be contiguous [M, C] of src's dtype")
        rc = self.fn["gather_rows"](_ptr(src), c, _ptr(rows), rows.shape[0], _ptr(out),
                                    self.stream(src.device))
        self._check(rc, "gather_rows")
        return out

    def scatter_add_rows(self, src: torch.Tensor, rows: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
        self._chk(src, torch.float32, "src")
        self._chk(dst, torch.float32, "dst")
        self._chk(rows, torch.int32, "rows")
        rc = self.fn["scatter_add_rows"](_ptr(src), src.shape[1], _ptr(rows), src.shape[0], _ptr(dst),
                                         self.stream(src.device))
        self._check(rc, "scatter_add_rows")
        return dst

    def sem_ensemble(self, logits, rows, want_conf: bool = True):
        """One-pass semantic ensembling (include/pasco_hip.h ph_sem_ensemble): logits[i] fp32 [n_i, c], rows[i] int32
        [n_sites] -> (outs: list of m + 1 fp32 [n_sites, c] rows - per subnet softmax resampled, then their mean;
        confs: list of m + 1 fp32 [n_sites] row maxima, or None)."""
        m = len(logits)
        c = logits[0].shape[1]
        n = rows[0].shape[0]
        dev = logits[0].device
        d = SemEnsDesc()
        d.m, d.c, d.n_sites = m, c, n
        outs = [torch.empty((n, c), dtype=torch.float32, device=dev) for _ in range(m + 1)]
        confs = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(m + 1)] if want_conf else None
        for i in range(m):
            self._chk(logits[i], torch.float32, "logits")
            self._chk(rows[i], torch.int32, "rows")
            if logits[i].shape[1] != c or rows[i].shape[0] != n:
                raise ValueError("sem_ensemble: shape mismatch between subnets")
            d.logits[i], d.rows[i] = _ptr(logits[i]), _ptr(rows[i])
        for i in range(m + 1):
            d.out[i] = _ptr(outs[i])
            d.conf[i] = _ptr(confs[i]) if want_conf else None
        rc = self.fn["sem_ensemble"](C.byref(d), self.stream(dev))
        self._check(rc, "sem_ensemble")
        return outs, confs

    # -- dense <-> sparse --------------------------------------------------------------------------
    def to_dense(self, feats, coords, min3, ts: int, dims4) -> torch.Tensor:
        self._chk(feats, torch.float32, "feats")
        self._chk(coords, torch.int32, "coords")
        c = feats.shape[1]
        b, x, y, z = [int(v) for v in dims4]
        dense = torch.zeros((b, c, x, y, z), dtype=torch.float32, device=feats.device)
        hmin = (_i32 * 3)(*[int(v) for v in min3])
        hdim = (_i32 * 4)(b, x, y, z)
        rc = self.fn["to_dense"](_ptr(feats), _ptr(coords), feats.shape[0], c, C.cast(hmin, _vp), int(ts),
                                 C.cast(hdim, _vp), _ptr(dense), self.stream(feats.device))
        self._check(rc, "to_dense")
        return dense

    def to_sparse(self, dense: torch.Tensor):
        """dense [B,C,X,Y,Z] -> (coords int32 [N,4] in site units, feats [N,C])."""
        self._chk(dense, torch.float32, "dense")
        b, c, x, y, z = dense.shape
        dev = dense.device
        nsites = b * x * y * z
        coords = torch.empty((max(nsites, 1), 4), dtype=torch.int32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        ws = self.workspace(nsites, dev)
        hdim = (_i32 * 4)(b, x, y, z)
        rc = self.fn["to_sparse_coords"](_ptr(dense), c, C.cast(hdim, _vp), _ptr(coords), _ptr(cnt),
                                         _ptr(ws), ws.numel(), self.stream(dev))
        self._check(rc, "to_sparse_coords")
        n = int(cnt.item())
        coords = coords[:n].contiguous()
        feats = self.dense_gather(dense, coords)
        return coords, feats

    def dense_gather(self, dense: torch.Tensor, site_coords: torch.Tensor) -> torch.Tensor:
        self._chk(dense, torch.float32, "dense")
        self._chk(site_coords, torch.int32, "site_coords")
        b, c, x, y, z = dense.shape
        n = site_coords.shape[0]
        feats = torch.empty((n, c), dtype=torch.float32, device=dense.device)
        hdim = (_i32 * 4)(b, x, y, z)
        rc = self.fn["dense_gather"](_ptr(dense), c, C.cast(hdim, _vp), _ptr(site_coords), n, _ptr(feats),
                                     self.stream(dense.device))
        self._check(rc, "dense_gather")
        return feats


    def keep_mask(self, srcs, coords: Optional[torch.Tensor] = None, lo: Optional[torch.Tensor] = None,
                  hi: Optional[torch.Tensor] = None, fallback_rows: int = 0) -> torch.Tensor:
        """bool [n]: OR over `srcs` (all bool / uint8 "kept" masks, or all int32 row tensors where >= 0 means kept - what
        `map_find` returns) AND the inclusive box test lo <= coords[:, 1:4] <= hi (device int32 [3] each, or both None);
        fallback_rows > 0: when nothing is kept at all, the first `fallback_rows` rows count as kept instead (decided on the
        device).  One pass (include/pasco_hip.h keep_mask) instead of a dozen element-wise torch kernels per mask."""
        srcs = list(srcs)
        if not srcs:                 # the box test alone
            if coords is None or lo is None:
                raise ValueError("keep_mask: no source and no box")
            kind, n, dev = 0, coords.shape[0], coords.device
        else:
            kind = 1 if srcs[0].dtype == torch.int32 else 0
            n = srcs[0].shape[0]
            dev = srcs[0].device
        for t in srcs:
            if t.shape != (n,) or not t.is_contiguous() or (t.dtype == torch.int32) != (kind == 1) or \
                    (kind == 0 and t.dtype not in (torch.bool, torch.uint8)):
                raise ValueError("keep_mask: sources must be contiguous [n] tensors, all int32 or all bool / uint8")
        if (lo is None) != (hi is None):
            raise ValueError("keep_mask: give both corners or neither")
        if lo is not None:
            self._chk(coords, torch.int32, "coords")
            self._chk(lo, torch.int32, "lo")
            self._chk(hi, torch.int32, "hi")
            if coords.shape != (n, 4) or lo.numel() != 3 or hi.numel() != 3:
                raise ValueError("keep_mask: coords [n, 4], corners [3]")
        out = torch.empty((n,), dtype=torch.bool, device=dev)
        ptrs = (_vp * len(srcs))(*[_ptr(t) for t in srcs])
        word = None
        if fallback_rows > 0:
            key = ("keep_any",) + self._stream_key(dev)
            word = self._ws.get(key)
            if word is None:
                word = torch.zeros(1, dtype=torch.int32, device=dev)
                self._ws[key] = word
        rc = self.fn["keep_mask"](C.cast(ptrs, _vp), len(srcs), kind, _ptr(coords) if lo is not None else None, n, _ptr(lo),
                                  _ptr(hi), int(fallback_rows), _ptr(out), _ptr(word), self.stream(dev))
        self._check(rc, "keep_mask")
        return out

    # -- attention -----------------------------------------------------------------------------------
    def attn_mask_pack(self, vals: torch.Tensor, b: int, n: int, positive_only: bool = False, want_any: bool = True):
        """[B*N, Qn] fp32 -> (bits int32 [B, N, 4], any int32 [B, 4] | None); a bit is set where the value
        is non-zero (or > 0 with positive_only)."""
        self._chk(vals, torch.float32, "vals")
        qn = vals.shape[1]
        assert vals.shape[0] == b * n
        bits = torch.empty((b, n, 4), dtype=torch.int32, device=vals.device)
        any_ = torch.empty((b, 4), dtype=torch.int32, device=vals.device) if want_any else None
        rc = self.fn["attn_mask_pack"](_ptr(vals), n, b, qn, 1 if positive_only else 0, _ptr(bits), _ptr(any_),
                                       self.stream(vals.device))
        self._check(rc, "attn_mask_pack")
        return bits, any_

    def bits_orpool(self, bits: torch.Tensor, nbr: torch.Tensor) -> torch.Tensor:
        """bits int32 [N_in, 4], nbr [K, N_out] -> OR over the neighbours [N_out, 4]."""
        self._chk(bits, torch.int32, "bits")
        self._chk(nbr, torch.int32, "nbr")
        kvol, n_out = nbr.shape
        out = torch.empty((n_out, 4), dtype=torch.int32, device=bits.device)
        rc = self.fn["bits_orpool"](_ptr(bits), _ptr(nbr), kvol, n_out, _ptr(out), self.stream(bits.device))
        self._check(rc, "bits_orpool")
        return out

    def bits_block_or(self, level_coords: torch.Tensor, n_per_b: int, s: int, tkeys, tvals, bits: torch.Tensor, lo=None,
                      hi=None, want_range: bool = False, range_word: Optional[torch.Tensor] = None):
        """level_coords int32 [M, 4] (batch = row // n_per_b), fine-map table (tkeys, tvals) and its bit rows [N1, 4] ->
        bits of the level voxels' s^3 blocks [M, 4] (+ a device flag word: some coordinate outside [lo, hi])."""
        self._chk(level_coords, torch.int32, "level_coords")
        self._chk(bits, torch.int32, "bits")
        m = level_coords.shape[0]
        out = torch.empty((m, 4), dtype=torch.int32, device=bits.device)
        rng = torch.zeros(1, dtype=torch.int32, device=bits.device) if want_range else range_word   # caller's word: OR-ed into
        if rng is not None:
            self._chk(lo, torch.int32, "lo")
            self._chk(hi, torch.int32, "hi")
        rc = self.fn["bits_block_or"](_ptr(level_coords), m, int(n_per_b), int(s), _ptr(tkeys), _ptr(tvals), tkeys.numel(),
                                      _ptr(bits), _ptr(lo), _ptr(hi), _ptr(out), _ptr(rng), self.stream(bits.device))
        self._check(rc, "bits_block_or")
        return (out, rng) if want_range else out

    def bits_or_reduce(self, bits: torch.Tensor) -> torch.Tensor:
        """bits int32 [B, N, 4] -> OR over N: [B, 4]."""
        self._chk(bits, torch.int32, "bits")
        b, n, _ = bits.shape
        out = torch.empty((b, 4), dtype=torch.int32, device=bits.device)
        rc = self.fn["bits_or_reduce"](_ptr(bits), n, b, _ptr(out), self.stream(bits.device))
        self._check(rc, "bits_or_reduce")
        return out

    def sine_pe(self, coords: torch.Tensor, dim_t: torch.Tensor, scale: float, coff: int = 0,
                table: Optional[torch.Tensor] = None, tab_lo: int = 0) -> torch.Tensor:
        """coords int32 [N, cstride] (x,y,z from column `coff`) -> [N, 3*f] sine position encoding.
        `table` [T, f] = `sine_pe_table(...)`: lookup for coordinate values tab_lo .. tab_lo + T - 1."""
        self._chk(coords, torch.int32, "coords")
        self._chk(dim_t, torch.float32, "dim_t")
        n, cstride = coords.shape
        f = dim_t.numel()
        tab_n = 0
        if table is not None:
            self._chk(table, torch.float32, "table")
            if table.dim() != 2 or table.shape[1] != f:
                raise ValueError("sine_pe: table must be [T, f]")
            tab_n = table.shape[0]
        out = torch.empty((n, 3 * f), dtype=torch.float32, device=coords.device)
        rc = self.fn["sine_pe"](_ptr(coords), n, cstride, coff, f, _ptr(dim_t), float(scale), _ptr(table), int(tab_lo),
                                tab_n, _ptr(out), self.stream(coords.device))
        self._check(rc, "sine_pe")
        return out

    def sine_pe_table(self, dim_t: torch.Tensor, scale: float, lo: int, hi: int) -> torch.Tensor:
        """[hi - lo, f] encodings of one axis for the integer values lo .. hi - 1 (evaluated by the same entry
        point, so a lookup returns exactly what the evaluation would)."""
        v = torch.arange(lo, hi, dtype=torch.int32, device=dim_t.device)
        c = torch.stack([v, v, v], dim=1).contiguous()
        return self.sine_pe(c, dim_t, scale)[:, : dim_t.numel()].contiguous()

    def attn_supported(self, qn: int, dh: int) -> bool:
        return qn <= 128 and (dh == 48 or self.device_type == "cpu")

    def attn_cross_fwd(self, q, k, v, bits=None, any_=None) -> torch.Tensor:
        """q [B,H,Qn,Dh] (pre-scaled), k/v [B,N,H*Dh] -> out [B,Qn,H*Dh]."""
        for t, nm in ((q, "q"), (k, "k"), (v, "v")):
            self._chk(t, torch.float32, nm)
        b, h, qn, dh = q.shape
        n = k.shape[1]
        assert k.shape == (b, n, h * dh) and v.shape == k.shape
        out = torch.empty((b, qn, h * dh), dtype=torch.float32, device=q.device)
        need = int(self.fn["attn_workspace_bytes"](n, b, h, qn, dh))
        key = ("attn",) + self._stream_key(q.device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q.device)
            self._ws[key] = ws
        rc = self.fn["attn_cross_fwd"](_ptr(q), _ptr(k), _ptr(v), _ptr(bits), _ptr(any_), _ptr(out), n, b, h, qn, dh,
                                       _ptr(ws), ws.numel(), self.stream(q.device))
        self._check(rc, "attn_cross_fwd")
        return out


    def attn_cross_split(self, q, k_split, v_split, n: int, bits=None, any_=None, exp2=None) -> torch.Tensor:
        """attn_cross_fwd on split f16 K / V operands ([B*N, H*Dh/32, 2, 32] f16 each, value * 2^exp2; what
        conv_fwd(..., out_split=) wrote): q [B,H,Qn,Dh] (pre-scaled) -> out [B,Qn,H*Dh]."""
        self._chk(q, torch.float32, "q")
        b, h, qn, dh = q.shape
        d = h * dh
        for t, nm in ((k_split, "k_split"), (v_split, "v_split")):
            assert t.dtype == torch.float16 and t.is_contiguous() and t.numel() == b * n * d * 2, nm
        exp2 = SPLIT_ACT_EXP2 if exp2 is None else int(exp2)
        out = torch.empty((b, qn, d), dtype=torch.float32, device=q.device)
        need = int(self.fn["attn_workspace_bytes"](n, b, h, qn, dh))
        key = ("attn",) + self._stream_key(q.device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q.device)
            self._ws[key] = ws
        rc = self.fn["attn_cross_split"](_ptr(q), _ptr(k_split), _ptr(v_split), exp2, _ptr(bits), _ptr(any_), _ptr(out),
                                         n, b, h, qn, dh, _ptr(ws), ws.numel(), self.status_ptr(q.device),
                                         self.stream(q.device))
        self._check(rc, "attn_cross_split")
        return out


    POS_AUG_COLS = 16

    def pos_aug(self, coords: torch.Tensor, eps: torch.Tensor, tab_lo: int, exp2: Optional[int] = None) -> torch.Tensor:
        """coords int32 [N, 4] -> position columns f16 [N, 16] of the keys (include/pasco_hip.h pos_aug): [c == 0] and
        eps[c - tab_lo] per axis, times 2^exp2 (default SPLIT_ACT_EXP2: the scale of the feature operand they extend)."""
        exp2 = SPLIT_ACT_EXP2 if exp2 is None else int(exp2)
        self._chk(coords, torch.int32, "coords")
        self._chk(eps, torch.float32, "eps")
        n = coords.shape[0]
        assert coords.shape[1] == 4
        out = torch.empty((n, self.POS_AUG_COLS), dtype=torch.float16, device=coords.device)
        rc = self.fn["pos_aug"](_ptr(coords), n, _ptr(eps), int(tab_lo), eps.numel(), exp2, _ptr(out),
                                self.status_ptr(coords.device), self.stream(coords.device))
        self._check(rc, "pos_aug")
        return out

    def attn_feat_supported(self, qn: int, c: int) -> bool:
        return qn <= 128 and (c == 64 or (self.device_type == "cpu" and c % 32 == 0)) and self.has("attn_cross_feat")

    def attn_cross_feat(self, q2, x_split, aug, n: int, bits=None, any_=None, exp2=None) -> torch.Tensor:
        """Attention on the level's feature operand (include/pasco_hip.h attn_cross_feat): q2 [B, H, Qn, c + 16],
        x_split f16 [B*N, c/32, 2, 32], aug f16 [B*N, 16] -> Y [B, Qn, H * (c + 16)]."""
        self._chk(q2, torch.float32, "q2")
        b, h, qn, d = q2.shape
        c = d - self.POS_AUG_COLS
        assert x_split.dtype == torch.float16 and x_split.is_contiguous() and x_split.numel() == b * n * c * 2, "x_split"
        assert aug.dtype == torch.float16 and aug.is_contiguous() and aug.numel() == b * n * self.POS_AUG_COLS, "aug"
        exp2 = SPLIT_ACT_EXP2 if exp2 is None else int(exp2)
        out = torch.empty((b, qn, h * d), dtype=torch.float32, device=q2.device)
        need = int(self.fn["attn_workspace_bytes"](n, b, h, qn, d))
        key = ("attn",) + self._stream_key(q2.device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=q2.device)
            self._ws[key] = ws
        rc = self.fn["attn_cross_feat"](_ptr(q2), _ptr(x_split), _ptr(aug), c, exp2, _ptr(bits), _ptr(any_), _ptr(out), n, b, h,
                                        qn, _ptr(ws), ws.numel(), self.status_ptr(q2.device), self.stream(q2.device))
        self._check(rc, "attn_cross_feat")
        return out

    # ---- panoptic ensembling rows (include/pasco_hip.h ens_*) ----------------------------------------------------------
    def project_canonical(self, T: torch.Tensor, size, resolution: float, min_bound) -> torch.Tensor:
        """-> int32 [X*Y*Z, 4] = (0, voxel index of T applied to the centre of every site of the size grid)."""
        T = T.to(dtype=torch.float32).contiguous()
        if T.device.type != self.device_type:
            raise ValueError(f"T on {T.device}, backend serves {self.device_type}")
        assert T.shape[-1] == 4 and T.shape[0] >= 3
        x, y, z = (int(v) for v in size)
        out = torch.empty((x * y * z, 4), dtype=torch.int32, device=T.device)
        mb = (C.c_float * 3)(*[float(v) for v in min_bound])
        rc = self.fn["project_canonical"](_ptr(T), x, y, z, float(resolution), C.cast(mb, _vp), _ptr(out), self.stream(T.device))
        self._check(rc, "project_canonical")
        return out

    def ens_resample(self, logits: torch.Tensor, rows: torch.Tensor, sel: torch.Tensor):
        """-> (probs [U, Q], flag uint8 [U]): sigmoid of the subnet's voxel logits resampled on the union sites `sel`
        (int32 canonical site ids) through `rows` (int32 [n_sites], -1 = no voxel -> zero row)."""
        self._chk(logits, torch.float32, "logits")
        self._chk(rows, torch.int32, "rows")
        self._chk(sel, torch.int32, "sel")
        n, q = logits.shape
        u = sel.shape[0]
        out = torch.empty((u, q), dtype=torch.float32, device=logits.device)
        flag = torch.empty((u,), dtype=torch.uint8, device=logits.device)
        rc = self.fn["ens_resample"](_ptr(logits), n, q, _ptr(rows), _ptr(sel), u, _ptr(out), _ptr(flag), self.stream(logits.device))
        self._check(rc, "ens_resample")
        return out, flag

    def ens_merge(self, anchor: torch.Tensor, m: torch.Tensor, perm: torch.Tensor, i: int) -> torch.Tensor:
        """anchor <- (anchor * i + m[:, perm]) / (i + 1), in place."""
        self._chk(anchor, torch.float32, "anchor")
        self._chk(m, torch.float32, "m")
        self._chk(perm, torch.int32, "perm")
        u, q = anchor.shape
        assert m.shape == anchor.shape and perm.numel() == q
        rc = self.fn["ens_merge"](_ptr(anchor), _ptr(m), _ptr(perm), u, q, int(i), self.stream(anchor.device))
        self._check(rc, "ens_merge")
        return anchor

    def ens_finish(self, anchor: torch.Tensor, keep: torch.Tensor, sem: torch.Tensor, sel: torch.Tensor):
        """-> (out [U, len(keep)], flag uint8 [U]): anchor's kept query columns, zeroed where the ensembled semantic class
        (argmax of sem[sel]) is 0."""
        self._chk(anchor, torch.float32, "anchor")
        self._chk(keep, torch.int32, "keep")
        self._chk(sem, torch.float32, "sem")
        self._chk(sel, torch.int32, "sel")
        u, q = anchor.shape
        qk = keep.numel()
        out = torch.empty((u, qk), dtype=torch.float32, device=anchor.device)
        flag = torch.empty((u,), dtype=torch.uint8, device=anchor.device)
        rc = self.fn["ens_finish"](_ptr(anchor), u, q, _ptr(keep), qk, _ptr(sem), sem.shape[1], _ptr(sel), _ptr(out), _ptr(flag),
                                   self.stream(anchor.device))
        self._check(rc, "ens_finish")
        return out, flag


    # ---- panoptic post-processing (include/pasco_hip.h panop_*) -----------------------------------------------------------
    PANOP_QMAX = 128

    def panoptic_rows(self, masks: torch.Tensor, qp: torch.Tensor, object_mask_threshold: float, overlap_threshold: float,
                      vox_occ_threshold: float, thing_ids, reduce_areas=None) -> dict:
        """`panoptic_inference` of one batch item on its sparse rows (helper.py:91-303), three launches and no host read:
        masks fp32 [n, q] mask probabilities, qp fp32 [q, c1] class probabilities -> device tensors {"panoptic" int32 [n],
        "semantic" int32 [n], "ins_unc", "vox_conf", "vox_unc" fp32 [n], "winner" int32 [n] (kept index or -1), "own" uint8
        [n], "qtab" int32 [4, 128], "nk" int32 [1], "seg" int32 [5, 128], "areas" int32 [2, 128], "tabs" int32 [12, 128] =
        the four tables in one tensor (rows 0-3 qtab, 4 nk, 5-9 seg, 10-11 areas: one copy brings them to the host)}.
        `reduce_areas(areas)`: called between the competition and the write pass - rows sharded over ranks (config C4,
        dist.site_sharded_panoptic) add their per-query areas there (exact integers), everything else is row-wise."""
        self._chk(masks, torch.float32, "masks")
        self._chk(qp, torch.float32, "qp")
        n, q = masks.shape
        c1 = qp.shape[1]
        if qp.shape[0] != q or not 1 <= q <= self.PANOP_QMAX or not 2 <= c1 <= 64:
            raise ValueError("panoptic_rows: masks [n, q], qp [q, c1] with q <= 128 queries and c1 <= 64 classes")
        dev = masks.device
        st = self.stream(dev)
        Q = self.PANOP_QMAX
        tabs = torch.zeros((4 + 1 + 5 + 2, Q), dtype=torch.int32, device=dev)        # qtab | nk | seg | areas: ONE fill
        qtab, nk, seg, areas = tabs[0:4], tabs[4, 0:1], tabs[5:10], tabs[10:12]
        self._check(self.fn["panop_queries"](_ptr(qp), q, c1, float(object_mask_threshold), qtab.data_ptr(), nk.data_ptr(),
                                             st), "panop_queries")
        per_i = torch.empty((3, max(n, 1)), dtype=torch.int32, device=dev)            # winner | panoptic | semantic
        per_f = torch.empty((5, max(n, 1)), dtype=torch.float32, device=dev)          # conf | vunc | ins_unc | vox_conf | vox_unc
        own = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        self._check(self.fn["panop_argmax"](_ptr(masks), n, q, qtab.data_ptr(), float(vox_occ_threshold), per_i[0].data_ptr(),
                                            _ptr(own), per_f[0].data_ptr(), per_f[1].data_ptr(), areas.data_ptr(), st),
                    "panop_argmax")
        if reduce_areas is not None:
            reduce_areas(areas)
        thing = 0
        for t in thing_ids:
            if 0 <= int(t) < 64:
                thing |= 1 << int(t)
        self._check(self.fn["panop_write"](n, per_i[0].data_ptr(), _ptr(own), per_f[0].data_ptr(), per_f[1].data_ptr(),
                                           areas.data_ptr(), qtab.data_ptr(), nk.data_ptr(), float(overlap_threshold), thing,
                                           per_i[1].data_ptr(), per_i[2].data_ptr(), per_f[2].data_ptr(), per_f[3].data_ptr(),
                                           per_f[4].data_ptr(), seg.data_ptr(), st), "panop_write")
        return {"panoptic": per_i[1, :n], "semantic": per_i[2, :n], "ins_unc": per_f[2, :n], "vox_conf": per_f[3, :n],
                "vox_unc": per_f[4, :n], "winner": per_i[0, :n], "own": own[:n], "qtab": qtab, "nk": nk, "seg": seg,
                "areas": areas, "tabs": tabs}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:      # older torch: the public (slower) route
    def _raw_stream(idx: int) -> int:
        return torch.cuda.current_stream(idx).cuda_stream


# ---- registry -----------------------------------------------------------------------------------
_hip_backend: Optional[CBackend] = None
_checker_backend: Optional[CBackend] = None


def hip_backend() -> CBackend:
    """The product backend. Raises if libpascohip.so is missing or cannot be loaded."""
    global _hip_backend
    if _hip_backend is None:
        _hip_backend = CBackend(HIP_LIB_PATH, "ph_", "cuda")
    return _hip_backend


def register_checker_backend(backend: Optional[CBackend]) -> None:
    """Test hook: serve CPU tensors from a checker library (the oracle). Never called by pasco_amd."""
    global _checker_backend
    _checker_backend = backend


def backend_for(device: torch.device) -> CBackend:
    device = torch.device(device)
    if device.type == "cuda":
        return hip_backend()
    if _checker_backend is not None and device.type == _checker_backend.device_type:
        return _checker_backend
    raise RuntimeError(
        f"pasco_amd serves ROCm GPU tensors only (got device '{device}'); there is no CPU path. "
        "Move tensors to cuda:N (MI355X) - tests may register a checker backend explicitly.")
