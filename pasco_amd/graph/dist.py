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

import time
from typing import Callable, List, Optional, Tuple

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
    for r in range(rounds):                      # rank k owns subnets k, k + world, ...
        have = r < len(local)
        if have:
            vl = local[r]["voxel_logits"]
            feats, coords, ql = vl.F.contiguous(), vl.C.contiguous(), local[r]["query_logits"].reshape(-1, n_cls)
        else:                                    # ragged tail: contribute empty rows
            feats = torch.zeros((0, n_q), device=dev)
            coords = torch.zeros((0, 4), dtype=torch.int32, device=dev)
            ql = torch.zeros((0, n_cls), device=dev)
        fs, cs = allgather_voxel_logits(feats, coords, group)
        qs = allgather_rows(ql.contiguous(), group)
        for k in range(world):
            i = k + r * world
            if i < net.n_infers and qs[k].shape[0]:
                full[i] = {"voxel_logits": ME.SparseTensor(fs[k], cs[k]), "query_logits": qs[k].reshape(1, n_q, n_cls),
                           "aux_outputs": local[r]["aux_outputs"] if (have and k == rank) else []}
    ret["panop_predictions"] = full
    return ret
