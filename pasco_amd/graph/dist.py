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


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership of independent scenes."""
    return list(range(rank, n_items, world))


def allgather_rows(rows: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather tensors that differ in their first dimension. Returns the list over ranks."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes) if sizes else 0
    pad = rows.new_zeros((n_max,) + tuple(rows.shape[1:]))
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
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
    dist.all_gather(all_rows, rows, group=group)
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
    dist.all_gather(box, mine, group=group)
    out = []
    for r in range(world):
        offs, _ = layout(counts[r])
        got = []
        for p, b, c, off in zip(parts, row_bytes, counts[r], offs):
            nb = c * b
            t = box[r][off:off + nb].view(p.dtype).view((c,) + tuple(p.shape[1:])) if nb else p.new_zeros((0,) + tuple(p.shape[1:]))
            got.append(t)
        out.append(got)
    return out, {"bytes_sent": int(payload), "bytes_padded": int(cap), "collectives": 2}


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
    stats = {"bytes_sent": 0, "bytes_padded": 0, "collectives": 0, "rounds": rounds, "payload": "f16" if half else "f32"}
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
        for k_ in ("bytes_sent", "bytes_padded", "collectives"):
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
