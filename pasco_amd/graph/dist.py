"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards two ways:
  * scene-parallel replicas - scenes are independent, rank r takes scenes r, r+W, ...; no data-path
    collective; this is what the scenes/sec metric at 1/2/4/8 GPUs measures (weak scaling);
  * subnet-parallel heads (config C4, MIMO M=8 with one subnet head per GPU) - one exchange step:
    an all-gather of the per-voxel logits so every rank can run the ensembler.  Row counts differ per
    subnet, so sizes are gathered first and payloads padded to the maximum (one all_gather each;
    on the xGMI mesh a direct all-gather moves every shard over its own link).
"""
from __future__ import annotations

import math
import os
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves host memory only (its device support is broadcast / all_reduce): device tensors of the other collectives are
    staged through the host.  RCCL ("nccl") takes device tensors directly - this branch is never taken on the product path."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(outs: List[torch.Tensor], t: torch.Tensor, group=None) -> None:
    if _host_staged(t, group):
        th = t.cpu()
        hs = [torch.empty_like(th) for _ in outs]
        dist.all_gather(hs, th, group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
        return
    dist.all_gather(outs, t, group=group)


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership of independent scenes."""
    return list(range(rank, n_items, world))


def allgather_rows(rows: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather tensors that differ in their first dimension. Returns the list over ranks."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    _all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes) if sizes else 0
    pad = rows.new_zeros((n_max,) + tuple(rows.shape[1:]))
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    _all_gather(out, pad.contiguous(), group=group)
    return [o[:s] for o, s in zip(out, sizes)]


def allgather_voxel_logits(feats: torch.Tensor, coords: torch.Tensor, group=None
                           ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """C4 exchange step: every rank contributes its subnet's per-voxel logits [N_i, C] and their
    coordinates [N_i, 4]; every rank receives all of them."""
    return allgather_rows(feats, group), allgather_rows(coords, group)


def packed_allgather(parts: Sequence[torch.Tensor], group=None) -> Tuple[List[List[torch.Tensor]], Dict[str, int]]:
    """One exchange step for SEVERAL ragged tensors (first dimensions differ per rank; trailing dimensions and dtypes are
    the same on every rank): ONE all-gather of the row counts ([world, len(parts)] int64, read once on the host) and ONE
    all-gather of a byte buffer holding every part back to back, padded to the longest rank.  Replaces 2 collectives + one
    host read PER PART (`allgather_rows`).  -> (out[rank][part], {"bytes_sent", "bytes_padded", "collectives"})."""
    world = dist.get_world_size(group)
    dev = parts[0].device
    rows = torch.tensor([int(p.shape[0]) for p in parts], dtype=torch.int64, device=dev)
    all_rows = [torch.zeros_like(rows) for _ in range(world)]
    _all_gather(all_rows, rows, group=group)
    counts = torch.stack(all_rows).tolist()                       # the one host read of the step
    row_bytes = [p.element_size() * math.prod(p.shape[1:]) for p in parts]
    ALIGN = 16                                                    # every part starts on a 16-byte boundary of the buffer: the
                                                                  # typed views below need offset % element_size == 0 whatever
                                                                  # the row counts and dtypes of the parts before it are

    def layout(cnt):                                              # byte offset of every part for one rank's row counts
        offs, off = [], 0
        for c, b in zip(cnt, row_bytes):
            offs.append(off)
            off = (off + c * b + ALIGN - 1) // ALIGN * ALIGN
        return offs, off

    sizes = [layout(cnt)[1] for cnt in counts]
    cap = max(max(sizes), ALIGN)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)        # padding between and behind the parts is zero
    offs, _ = layout([int(p.shape[0]) for p in parts])
    payload = 0
    for p, b, off in zip(parts, row_bytes, offs):
        nb = int(p.shape[0]) * b
        payload += nb
        if nb:
            mine[off:off + nb] = p.contiguous().view(-1).view(torch.uint8)
    box = [torch.empty_like(mine) for _ in range(world)]
    _all_gather(box, mine, group=group)
    out = []
    for r in range(world):
        offs, _ = layout(counts[r])
        got = []
        for p, b, c, off in zip(parts, row_bytes, counts[r], offs):
            nb = c * b
            t = box[r][off:off + nb].view(p.dtype).view((c,) + tuple(p.shape[1:])) if nb else p.new_zeros((0,) + tuple(p.shape[1:]))
            got.append(t)
        out.append(got)
    received = sum(sizes[r] for r in range(world)) - sizes[dist.get_rank(group)]
    return out, {"bytes_sent": int(payload), "bytes_padded": int(cap), "bytes_received": int(received), "collectives": 2}


def cpu_list(text: str) -> List[int]:
    """Linux cpulist syntax ("0-15,128-143") -> sorted CPU numbers."""
    out = []
    for piece in text.strip().split(","):
        if not piece:
            continue
        lo, _, hi = piece.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def _core_groups(cpus: Sequence[int], sysfs: str) -> List[List[int]]:
    """The CPUs grouped by physical core (SMT siblings together), cores in ascending order of their first CPU."""
    seen, groups = set(), []
    want = set(cpus)
    for c in sorted(cpus):
        if c in seen:
            continue
        sib = [c]
        for name in ("core_cpus_list", "thread_siblings_list"):
            try:
                with open(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", name)) as f:
                    sib = [v for v in cpu_list(f.read()) if v in want] or [c]
                break
            except OSError:
                continue
        seen.update(sib)
        groups.append(sorted(sib))
    return groups


def rank_cpu_set(local_rank: int, n_local: int, pci_bus_id: Optional[str] = None, sysfs: str = "/sys",
                 allowed: Optional[Sequence[int]] = None) -> List[int]:
    """CPUs rank `local_rank` of `n_local` ranks on this host should run on: whole physical cores (SMT siblings stay
    together) out of the CPUs LOCAL to its GPU's PCIe root (`/sys/bus/pci/devices/<id>/local_cpulist`; MI355X nodes hang
    4 GPUs off each socket), split evenly between the ranks of that socket (GPUs are enumerated socket by socket);
    without that information an even slice of the allowed cores.  A rank runs several Python threads (scenes in flight)
    that fight over the interpreter lock, plus OpenMP / torch helper threads: left unpinned, 8 ranks x 3 threads migrate
    across both sockets and their host-side launch latency - what bounds a scene here - degrades with the rank count."""
    allowed = sorted(allowed) if allowed is not None else sorted(os.sched_getaffinity(0))
    local = None
    if pci_bus_id:
        try:
            with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "local_cpulist")) as f:
                ok = set(allowed)
                local = [c for c in cpu_list(f.read()) if c in ok]
        except OSError:
            local = None
    if local and len(local) < len(allowed):
        domains = max(1, round(len(allowed) / len(local)))       # sockets / NUMA domains with GPUs
        ranks_here = max(1, -(-n_local // domains))              # ranks whose GPUs hang off this domain
        pos = local_rank % ranks_here
        cores = _core_groups(local, sysfs)
    else:
        ranks_here, pos = max(n_local, 1), local_rank % max(n_local, 1)
        cores = _core_groups(allowed, sysfs)
    per = max(1, len(cores) // ranks_here)
    mine = cores[pos * per:(pos + 1) * per] or cores
    return sorted(c for g in mine for c in g)


def pin_rank(local_rank: int, n_local: int, device_index: Optional[int] = None) -> List[int]:
    """Apply `rank_cpu_set` to the calling process (all its future threads inherit it) -> the CPUs chosen ([] when the
    platform has no sched_setaffinity or PASCO_BENCH_PIN=0)."""
    if os.environ.get("PASCO_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return []
    bus = None
    try:
        if device_index is not None and torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(device_index)
            if hasattr(pr, "pci_bus_id"):
                bus = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
    except Exception:
        bus = None
    cpus = rank_cpu_set(local_rank, n_local, bus)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return []
    return cpus


def timed_steps(step: Callable[[], None], steps: int, warmup: int, device_sync: Optional[Callable[[], None]] = None
                ) -> float:
    """bench.py's timing contract: W untimed warm-up steps, barrier + device sync, K timed steps,
    barrier + device sync, MAX over ranks of the elapsed seconds."""
    distributed = dist.is_available() and dist.is_initialized()

    def fence():
        if distributed:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def subnet_parallel_forward(net, in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs, keep_override=None,
                            group=None):
    """Config C4 (one MIMO head per GPU): every rank runs the shared trunk on the same scene, then only
    its own subnets' voxel-feature convolutions and transformer rows; one exchange step all-gathers
    the per-voxel mask logits (+ coordinates) and the query logits so that every rank can ensemble.
    Returns the same dict as `net(...)` with `panop_predictions` complete on every rank."""
    from .. import me as ME
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = shard_indices(net.n_infers, rank, world)
    ret = net(in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs, keep_override=keep_override,
              subnets=mine)
    local = ret["panop_predictions"]
    dev = in_feat.device
    n_q = net.transformer_predictor.num_queries
    n_cls = net.n_classes + 1
    full = [None] * net.n_infers
    rounds = (net.n_infers + world - 1) // world
    half = os.environ.get("PASCO_C4_EXCHANGE", "f32") == "f16"   # opt-in: mask logits travel as f16 (half the bytes, ~1e-3 rel)
    stats = {"bytes_sent": 0, "bytes_padded": 0, "bytes_received": 0, "collectives": 0, "rounds": rounds,
             "payload": "f16" if half else "f32"}
    for r in range(rounds):                      # rank k owns subnets k, k + world, ...
        have = r < len(local)
        if have:
            vl = local[r]["voxel_logits"]
            feats, coords, ql = vl.F.contiguous(), vl.C.contiguous(), local[r]["query_logits"].reshape(-1, n_cls)
        else:                                    # ragged tail: contribute empty rows
            feats = torch.zeros((0, n_q), device=dev)
            coords = torch.zeros((0, 4), dtype=torch.int32, device=dev)
            ql = torch.zeros((0, n_cls), device=dev)
        # ONE exchange: row counts, then one packed buffer (mask logits | coordinates | query logits) per rank
        got, st = packed_allgather([feats.half() if half else feats, coords, ql.contiguous()], group)
        for k_ in ("bytes_sent", "bytes_padded", "bytes_received", "collectives"):
            stats[k_] += st[k_]
        for k in range(world):
            i = k + r * world
            fs, cs, qs = got[k]
            if i < net.n_infers and qs.shape[0]:
                full[i] = {"voxel_logits": ME.SparseTensor(fs.float() if half else fs, cs),
                           "query_logits": qs.reshape(1, n_q, n_cls),
                           "aux_outputs": local[r]["aux_outputs"] if (have and k == rank) else []}
    ret["panop_predictions"] = full
    ret["exchange"] = stats
    return ret


# ---- config C4, site-sharded ensembling -------------------------------------------------------------------------------
# `subnet_parallel_forward` above ships every subnet's [N_i, Q] mask logits to every rank (all-gather: each rank receives
# (W - 1) / W of ALL masks, 7 x 84 MB at S10 / M = 8) and then every rank runs the whole ensembler redundantly.  Here the rows
# of the ensembling - the union of occupied canonical sites - are cut into GRAM_SLABS contiguous slabs owned by the ranks;
# a rank resamples ITS subnet's masks onto the union rows and sends every other rank only that rank's slab (all-to-all:
# each rank receives (W - 1) / W of ONE mask tensor in total, 8 x fewer bytes per link at W = 8).  Everything per row
# (running mean of the matched masks, class-0 zeroing, output compaction, the panoptic competition) is local to a slab; the
# only sums over rows - the soft-IoU matching's Gram matrix and column sums, the panoptic areas - travel as per-slab partial
# results ([Q * Q + 2 Q] floats per slab, 40 KB) and are added in SLAB ORDER on every rank: the same partials in the same
# order as `Ensembler.match_queries` adds them in one process, so the ensemble is bit-identical to the single-process one.

_SLAB_BUFFERS: Dict[tuple, torch.Tensor] = {}


def _persistent(key: tuple, shape, dtype, device) -> torch.Tensor:
    """One exchange buffer per (purpose, dtype, device), reused from step to step (no allocation in the timed step): a flat buffer
    GROWN to the largest size asked for so far (capacity rounded up to 1 / 8 of a power of two; the previous buffer is dropped
    when it grows) and viewed in the asked shape - a scene's slab row count differs from the last scene's nearly always, and a
    buffer per exact shape kept up to 64 dead arenas of 100 - 400 MB each."""
    k = key + (dtype, str(device))
    need = 1
    for v in shape:
        need *= int(v)
    t = _SLAB_BUFFERS.get(k)
    if t is None or t.numel() < need:
        step = max(1 << max(need.bit_length() - 4, 0), 1)
        cap = (need + step - 1) // step * step
        _SLAB_BUFFERS.pop(k, None)
        t = torch.empty(cap, dtype=dtype, device=device)
        _SLAB_BUFFERS[k] = t
    return t[:need].view(*shape)


def slab_owner(k: int, world: int, slabs: int) -> int:
    return k * world // slabs


def exchange_blocks(send: List[torch.Tensor], recv: List[torch.Tensor], group=None) -> None:
    """All-to-all of per-destination blocks (send[d] goes to rank d, recv[k] comes from rank k; sizes known on both
    sides).  RCCL: one `all_to_all`; gloo (CPU tests) has none: pairwise non-blocking send / receive."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(recv, send, group=group)
        return
    if recv[rank].numel():
        recv[rank].copy_(send[rank])
    staged = any(_host_staged(t, group) for t in send + recv)
    hsend = [t.cpu() if (staged and t.numel()) else t for t in send]
    hrecv = [torch.empty(t.shape, dtype=t.dtype) if (staged and t.numel()) else t for t in recv]
    ops = []
    for k in range(world):
        if k == rank:
            continue
        if send[k].numel():
            ops.append(dist.P2POp(dist.isend, hsend[k], k, group))
        if recv[k].numel():
            ops.append(dist.P2POp(dist.irecv, hrecv[k], k, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged:
        for k in range(world):
            if k != rank and recv[k].numel():
                recv[k].copy_(hrecv[k])


def site_sharded_ensemble(net, ret, Ts, group=None):
    """Ensembling of config C4 with the union rows sharded by slab.  `ret` = what `net(..., subnets=mine)` returned on this
    rank (`sem_logits_at_scales` complete - the completion heads are part of the shared trunk - and `panop_predictions` for
    the subnets this rank owns, `shard_indices(M, rank, W)` in order).  -> (sem_prob_denses, sharded, stats): `sharded` =
    one dict per output (the M subnets, then the ensemble): {"sites" int32 [n] canonical site ids of this rank's kept rows,
    "voxel_probs" [n, Q'], "sem_probs" [n, C], "query_probs"}; rows of rank 0, 1, ... concatenated are the single-process
    `Ensembler.ensemble_panop` outputs (`gather_sharded` does that).
    One scene at a time per process: the receive arena is persistent and shared by successive calls (config C4 issues its
    collectives from one thread, `bench.py --mode subnet-heads` sets `in_flight` to 1)."""
    import torch.nn.functional as F
    from ..me.backend import backend_for
    from . import ensemble as E
    ens = net.ensembler
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    M = net.n_infers
    mine = shard_indices(M, rank, world)
    local = ret["panop_predictions"]
    assert len(local) == len(mine), "ret must come from net(..., subnets=shard_indices(M, rank, world))"
    dev = ret["sem_logits_at_scales"][1][0].F.device
    be = backend_for(dev)
    S = E.GRAM_SLABS
    assert world <= S, f"at most {S} ranks (one slab of rows each)"
    stats = {"bytes_sent": 0, "bytes_received": 0, "collectives": 0, "payload": "f32 mask probabilities, slab rows only"}
    cache = {}
    sem_prob_denses = ens.ensemble_sem_compl(ret["sem_logits_at_scales"], Ts, cache=cache)     # replicated (one pass)
    sem_rows = cache["sem_rows"]
    n_sites = sem_rows[0].shape[0]
    # 1. occupancy of the union: OR over all subnets' lookups, one small all-reduce (1 byte per canonical site)
    rows_of = {}
    occ = torch.zeros(n_sites, dtype=torch.uint8, device=dev)
    for r, i in enumerate(mine):
        rows_of[i] = E._lookup_rows(local[r]["voxel_logits"], ens.projected(Ts[i], dev, cache))
        occ |= (rows_of[i] >= 0).to(torch.uint8)
    dist.all_reduce(occ, op=dist.ReduceOp.MAX, group=group)
    stats["bytes_sent"] += int(occ.numel())
    stats["bytes_received"] += int(occ.numel())
    stats["collectives"] += 1
    union_sites = be.mask_compact(occ.contiguous())                   # identical on every rank
    U = int(union_sites.shape[0])
    b = E.slab_bounds(U, S)
    owner = [slab_owner(k, world, S) for k in range(S)]
    my_slabs = [k for k in range(S) if owner[k] == rank]
    row_lo = {d: b[min(k for k in range(S) if owner[k] == d)] for d in range(world) if d in owner}
    row_hi = {d: b[max(k for k in range(S) if owner[k] == d) + 1] for d in range(world) if d in owner}
    lo, hi = (row_lo[rank], row_hi[rank]) if my_slabs else (0, 0)
    R = hi - lo
    nq = net.transformer_predictor.num_queries
    n_cls1 = net.n_classes + 1
    # 2. every subnet's class probabilities of the queries on every rank (tiny)
    rounds = (M + world - 1) // world
    qmine = torch.zeros((rounds, nq, n_cls1), dtype=torch.float32, device=dev)
    for r in range(len(mine)):
        qmine[r] = F.softmax(local[r]["query_logits"].reshape(nq, n_cls1), dim=-1)
    qall = [torch.empty_like(qmine) for _ in range(world)]
    _all_gather(qall, qmine, group=group)
    stats["bytes_sent"] += int(qmine.numel() * 4)
    stats["bytes_received"] += int(qmine.numel() * 4 * (world - 1))
    stats["collectives"] += 1
    query_probs = [qall[i % world][i // world].reshape(1, nq, n_cls1) for i in range(M)]
    # 3. masks: resample the own subnet on the union rows, keep the own slab, send every other rank its slab
    rowk = be.has("ens_resample") and nq <= E.ENS_KERNEL_MAX_Q and sem_rows[-1].shape[1] <= E.ENS_KERNEL_MAX_C
    ens_resample = be.ens_resample if rowk else E._ens_resample_torch
    ens_merge = be.ens_merge if rowk else E._ens_merge_torch
    ens_finish = be.ens_finish if rowk else E._ens_finish_torch
    arena = _persistent(("c4-masks", M), (M, max(R, 1), nq), torch.float32, dev)     # this rank's slab of every subnet
    masks = [None] * M
    empty = torch.empty((0, nq), dtype=torch.float32, device=dev)
    for r in range(rounds):
        send = [empty] * world
        if r < len(mine):
            i = mine[r]
            m, _ = ens_resample(local[r]["voxel_logits"].F.contiguous(), rows_of[i].contiguous(), union_sites)
            send = [m[row_lo[d]:row_hi[d]] if d in row_lo else empty for d in range(world)]
            stats["bytes_sent"] += int(sum(t.numel() for d, t in enumerate(send) if d != rank) * 4)
        recv = []
        for k in range(world):
            i = k + r * world
            recv.append(arena[i, :R] if (i < M and R > 0) else empty)
        # a rank without rows (more ranks than slabs with rows) neither receives nor is sent anything
        send = [t.contiguous() if (d in row_lo and row_hi[d] > row_lo[d]) else empty for d, t in enumerate(send)]
        exchange_blocks(send, recv, group)
        stats["collectives"] += 1
        stats["bytes_received"] += int(sum(t.numel() for k, t in enumerate(recv) if k != rank) * 4)
        for k in range(world):
            i = k + r * world
            if i < M:
                masks[i] = arena[i, :R]
    flags = [(m != 0).any(dim=1).to(torch.uint8) for m in masks]
    # 4. running mean of the matched masks; the matching's sums over the rows travel as per-slab partials
    local_b = [b[k] - lo for k in my_slabs] + ([b[my_slabs[-1] + 1] - lo] if my_slabs else [])
    per_rank = max(sum(1 for k in range(S) if owner[k] == d) for d in range(world))
    P = nq * nq + 2 * nq
    anchor_q = query_probs[0].clone()
    anchor_m = masks[0].clone() if M > 1 else masks[0]
    ious = []
    for i in range(1, M):
        part = torch.zeros((per_rank, P), dtype=torch.float32, device=dev)
        for j in range(len(my_slabs)):
            part[j] = E.match_partials(anchor_m[local_b[j]:local_b[j + 1]], masks[i][local_b[j]:local_b[j + 1]])
        allp = [torch.empty_like(part) for _ in range(world)]
        _all_gather(allp, part, group=group)
        stats["bytes_sent"] += int(part.numel() * 4)
        stats["bytes_received"] += int(part.numel() * 4 * (world - 1))
        stats["collectives"] += 1
        seen = [0] * world
        parts = []
        for k in range(S):                                      # slab order, whichever rank computed the slab
            parts.append(allp[owner[k]][seen[owner[k]]])
            seen[owner[k]] += 1
        a_idx, b_idx, iou = E.match_from_partials(parts, nq, net.iou_threshold)
        anchor_q = (anchor_q * i + query_probs[i][:, b_idx, :]) / (i + 1)
        if R > 0:
            ens_merge(anchor_m, masks[i], b_idx.to(torch.int32).contiguous(), i)
        ious.append(iou)
    if ious:
        keep_cols = (torch.stack(ious, dim=0).mean(0) > net.iou_threshold).nonzero().reshape(-1).to(dev)
        anchor_q = anchor_q.index_select(1, keep_cols)
    else:
        keep_cols = torch.arange(nq, device=dev)
    my_sites = union_sites[lo:hi].contiguous()
    if R > 0:
        ens_m, ens_flag = ens_finish(anchor_m.contiguous(), keep_cols.to(torch.int32).contiguous(), sem_rows[-1].contiguous(), my_sites)
    else:
        ens_m, ens_flag = torch.empty((0, int(keep_cols.numel())), device=dev), torch.empty(0, dtype=torch.uint8, device=dev)
    masks.append(ens_m)
    flags.append(ens_flag)
    query_probs.append(anchor_q)
    # 5. outputs: the non-zero rows of this rank's slab (ME.to_sparse keeps non-zero sites)
    nz_rows = be.mask_compact_many(flags) if R > 0 else [torch.empty(0, dtype=torch.int32, device=dev) for _ in flags]
    sharded = []
    for i, m in enumerate(masks):
        nz = nz_rows[i]
        sites = be.gather_rows(my_sites.reshape(-1, 1), nz).reshape(-1) if nz.numel() else my_sites[:0]
        vf = be.gather_rows(m.contiguous(), nz) if (m.shape[1] > 0 and nz.numel()) else m.new_zeros((int(nz.shape[0]), m.shape[1]))
        sf = be.gather_rows(sem_rows[i], sites) if sites.numel() else sem_rows[i][:0]
        sharded.append({"sites": sites, "voxel_probs": vf, "sem_probs": sf, "query_probs": query_probs[i]})
    stats["MB_sent_per_rank"] = round(stats["bytes_sent"] / 1e6, 3)
    stats["MB_received_per_rank"] = round(stats["bytes_received"] / 1e6, 3)
    stats["rows"] = {"union": U, "mine": R}
    return sem_prob_denses, sharded, stats


def gather_sharded(net, sharded, group=None):
    """The sharded outputs of `site_sharded_ensemble` as complete SparseTensors on every rank (tests, consumers that need
    the masks in one place; the per-voxel label maps of `site_sharded_panoptic` are what normally travels)."""
    from .. import me as ME
    from ..me.backend import backend_for
    out = []
    sites_grid = None
    for o in sharded:
        dev = o["voxel_probs"].device
        sites = torch.cat(allgather_rows(o["sites"].reshape(-1, 1), group)).reshape(-1)
        vp = torch.cat(allgather_rows(o["voxel_probs"], group))
        sp = torch.cat(allgather_rows(o["sem_probs"], group))
        if sites_grid is None:
            sites_grid = net.ensembler.sites(dev)
        xyz = sites_grid.index_select(0, sites.long())
        coords = torch.cat([torch.zeros((xyz.shape[0], 1), dtype=torch.int32, device=dev), xyz], dim=1).contiguous()
        mgr = ME.CoordinateManager(D=3, device=dev)
        key = mgr.insert_unique(coords, 1)
        out.append({"voxel_probs": ME.SparseTensor(vp, coordinate_map_key=key, coordinate_manager=mgr),
                    "sem_probs": ME.SparseTensor(sp, coordinate_map_key=key, coordinate_manager=mgr),
                    "query_probs": o["query_probs"]})
    return out


def site_sharded_panoptic(net, sharded, group=None):
    """`panoptic_inference` (helper.py:91-303) of every sharded output on this rank's rows: the per-voxel competition is
    local, the per-query areas are added over the ranks (exact integers, one 1 KB all-reduce per output), the walk over the
    kept queries then gives the same segments on every rank.  -> one dict per output: {"sites", "panoptic", "semantic",
    "ins_unc", "vox_conf", "vox_unc" (this rank's rows), "segments_infos"}."""
    from ..me.backend import backend_for
    outs = []
    for o in sharded:
        masks = o["voxel_probs"].contiguous()
        dev = masks.device
        be = backend_for(dev)
        qp = o["query_probs"][0].contiguous().float()

        def reduce_areas(areas):
            dist.all_reduce(areas, op=dist.ReduceOp.SUM, group=group)
        if qp.shape[0] == 0:             # no query survived the matching filter: nothing to compete
            n = masks.shape[0]
            z = lambda dt: torch.zeros(n, dtype=dt, device=dev)
            outs.append({"sites": o["sites"], "panoptic": z(torch.int32), "semantic": z(torch.int32), "ins_unc": z(torch.float32),
                         "vox_conf": z(torch.float32), "vox_unc": z(torch.float32), "segments_infos": []})
            continue
        rows = be.panoptic_rows(masks, qp, net.object_mask_threshold, net.overlap_threshold, 0.3, net.thing_ids,
                                reduce_areas=reduce_areas)
        t = rows["tabs"].cpu()
        probs = t[3].view(torch.float32)
        infos = [{"id": int(t[5, s]), "isthing": bool(t[6, s]), "category_id": int(t[7, s]), "query_id": int(t[8, s]),
                  "confidence": float(probs[int(t[8, s])]), "all_class_probs": qp[int(t[8, s])]} for s in range(int(t[9, 0]))]
        outs.append({"sites": o["sites"], "panoptic": rows["panoptic"], "semantic": rows["semantic"], "ins_unc": rows["ins_unc"],
                     "vox_conf": rows["vox_conf"], "vox_unc": rows["vox_unc"], "segments_infos": infos})
    return outs


def subnet_parallel_local(net, in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs, keep_override=None, group=None):
    """The compute half of config C4 without any exchange: the shared trunk and this rank's subnet heads -> the dict
    `site_sharded_ensemble` takes (`panop_predictions` = the subnets `shard_indices(M, rank, W)` in order)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    return net(in_feat, global_min_coords, global_max_coords, min_Cs, max_Cs, keep_override=keep_override,
               subnets=shard_indices(net.n_infers, rank, world))
