"""N > 1 path on CPU: world_size-2 gloo processes exercise scene sharding, the max-over-ranks timing
contract of bench.py and the variable-size all-gather of per-voxel logits (config C4)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pasco_amd.graph.dist import allgather_voxel_logits, packed_allgather, rank_cpu_set, shard_indices, timed_steps


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        n = 50 + 37 * rank                       # subnets keep different numbers of voxels
        feats = torch.randn(n, 20, generator=g)
        coords = torch.randint(0, 256, (n, 4), generator=g).int()
        fs, cs = allgather_voxel_logits(feats, coords)
        ok = len(fs) == world
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            ef = torch.randn(50 + 37 * r, 20, generator=gr)
            ec = torch.randint(0, 256, (50 + 37 * r, 4), generator=gr).int()
            ok = ok and torch.equal(fs[r], ef) and torch.equal(cs[r], ec)
        # the packed form of the same exchange: two collectives for all parts, bytes accounted
        ql = torch.randn(3 * rank, 21, generator=g)              # rank 0 contributes an EMPTY part
        got, st = packed_allgather([feats, coords, ql])
        ok = ok and st["collectives"] == 2 and st["bytes_sent"] == n * 20 * 4 + n * 4 * 4 + 3 * rank * 21 * 4
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            ef = torch.randn(50 + 37 * r, 20, generator=gr)
            ec = torch.randint(0, 256, (50 + 37 * r, 4), generator=gr).int()
            eq = torch.randn(3 * r, 21, generator=gr)
            ok = ok and torch.equal(got[r][0], ef) and torch.equal(got[r][1], ec) and torch.equal(got[r][2], eq)
            ok = ok and got[r][1].dtype == torch.int32 and tuple(got[r][2].shape) == (3 * r, 21)
        # timing contract: the slow rank sets the time for everybody
        import time
        elapsed = timed_steps(lambda: time.sleep(0.01 * (1 + 3 * rank)), steps=3, warmup=1)
        mine = shard_indices(7, rank, world)
        q.put((rank, ok, elapsed, mine))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "all-gather of voxel logits"
    e0, e1 = res[0][2], res[1][2]
    assert abs(e0 - e1) < 1e-9 and e0 >= 3 * 0.04 * 0.9, "max over ranks"
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]


def _c4_worker(rank, world, port, q, n_infers=2):
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.conftest import load_oracle
        from pasco_amd.me import backend
        from pasco_amd.graph import PascoNet
        from pasco_amd.graph.dist import subnet_parallel_forward
        from pasco_amd.graph.synth import TeacherKeep, make_scene
        backend.register_checker_backend(load_oracle())
        torch.manual_seed(3)
        net = PascoNet(n_classes=20, n_infers=n_infers, in_channels=12, f=8, num_queries=8, heavy_decoder=False).eval()
        net.ensembler.scene_size = (24, 24, 8)
        sc = make_scene(4, n_infers=n_infers, in_channels=12, grid=(24, 24, 8), occupancy=0.12)
        tk = TeacherKeep(sc, "cpu")
        with torch.no_grad():
            x = net.prepare_input(sc.in_feats, sc.in_coords)
            args = (x, sc.global_min_Cs, sc.global_max_Cs, sc.min_Cs, sc.max_Cs)
            ref = net(*args, keep_override=tk)                                  # all heads on one process
            got = subnet_parallel_forward(net, *args, keep_override=tk)         # one head per rank + all-gather
            ok = len(got["panop_predictions"]) == n_infers and all(p is not None for p in got["panop_predictions"])
            ex = got["exchange"]
            ok = ok and ex["rounds"] == -(-n_infers // world) and ex["collectives"] == 2 * ex["rounds"] and ex["bytes_sent"] > 0
            for a, b in zip(got["panop_predictions"], ref["panop_predictions"]):
                ok = ok and torch.equal(a["voxel_logits"].C, b["voxel_logits"].C)
                ok = ok and torch.allclose(a["voxel_logits"].F, b["voxel_logits"].F, rtol=1e-4, atol=1e-5)
                ok = ok and torch.allclose(a["query_logits"], b["query_logits"], rtol=1e-4, atol=1e-5)
            e_got = net.ensemble(got, sc.Ts)[2]
            e_ref = net.ensemble(ref, sc.Ts)[2]
            ok = ok and torch.allclose(e_got[-1]["voxel_probs"].F, e_ref[-1]["voxel_probs"].F, rtol=1e-4, atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_subnet_parallel_heads_world2_gloo():
    """Config C4 in miniature: MIMO M=2, one subnet head per process, all-gather of per-voxel logits;
    every rank ends with the same predictions as the single-process graph."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)


def test_subnet_parallel_heads_ragged_tail_world4_gloo():
    """n_infers NOT divisible by the world size (M = 6 on 4 ranks: the second exchange round has two ranks with nothing to
    contribute): every rank still ends with all six subnets' predictions, equal to the single-process graph."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q, 6)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == world and all(r[1] for r in res)


def test_rank_cpu_sets_follow_the_gpu_numa_node(tmp_path):
    """8 ranks on a 2-socket host with SMT (fake sysfs): each rank gets whole physical cores of ITS GPU's socket, the sets
    of different ranks are disjoint, and without PCI information the allowed cores are split evenly."""
    sysfs = tmp_path
    ncpu = 32                                               # cores 0-15, SMT siblings 16-31; socket 0 = cores 0-7, socket 1 = 8-15
    for c in range(ncpu):
        d = sysfs / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        core = c % 16
        (d / "core_cpus_list").write_text(f"{core},{core + 16}\n")
    for g in range(8):
        d = sysfs / "bus" / "pci" / "devices" / f"0000:{g:02x}:00.0"
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text("0-7,16-23\n" if g < 4 else "8-15,24-31\n")
    sets = [rank_cpu_set(r, 8, f"0000:{r:02x}:00.0", sysfs=str(sysfs), allowed=range(ncpu)) for r in range(8)]
    assert sets[0] == [0, 1, 16, 17] and sets[3] == [6, 7, 22, 23] and sets[4] == [8, 9, 24, 25]
    flat = [c for s_ in sets for c in s_]
    assert len(flat) == len(set(flat)) == ncpu
    even = [rank_cpu_set(r, 4, None, sysfs=str(sysfs), allowed=range(ncpu)) for r in range(4)]
    assert even[1] == [4, 5, 6, 7, 20, 21, 22, 23] and sum(len(e) for e in even) == ncpu
    one = rank_cpu_set(0, 1, "0000:05:00.0", sysfs=str(sysfs), allowed=range(ncpu))
    assert one == [8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 28, 29, 30, 31]      # a single rank: its GPU's whole socket


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher re-executes itself under torch.distributed.run (one rank per
    GPU) - exercised here on CPU / gloo through the hidden --dry-run mode (a sleep instead of a scene): the JSON line
    carries the rank count the process group saw and the max-over-ranks time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--dry-run"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["n_ranks"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert len(r["per_rank_ms_per_step"]) == 2
    # rank 1 sleeps twice as long: the slow rank sets the time
    assert r["ms_per_step"] >= 0.9 * max(r["per_rank_ms_per_step"]) and r["ms_per_step"] >= 9.0
    # single process: the same path without a process group
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "0", "--dry-run"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_ranks"] == 1


# ---- config C4 with the ensembling sharded by canonical-site slab (dist.site_sharded_ensemble) ------------------------------
def _sharded_worker(rank, world, port, q, n_infers, device="cpu"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.conftest import load_oracle
        from pasco_amd.me import backend
        from pasco_amd.graph import PascoNet
        from pasco_amd.graph.dist import (allgather_rows, gather_sharded, shard_indices, site_sharded_ensemble,
                                          site_sharded_panoptic, subnet_parallel_forward)
        from pasco_amd.graph.synth import TeacherKeep, make_scene
        if device == "cpu":
            backend.register_checker_backend(load_oracle())
        torch.manual_seed(3)
        net = PascoNet(n_classes=20, n_infers=n_infers, in_channels=12, f=8, num_queries=8, heavy_decoder=False,
                       object_mask_threshold=0.05).eval().to(device)          # random queries: a low bar so that segments exist
        net.ensembler.scene_size = (24, 24, 8)
        from pasco_amd.graph.ensemble import GRAM_SLABS
        net.ensembler.gram_slabs = GRAM_SLABS              # the single-process reference adds the slabs in the sharded run's order
        sc = make_scene(4, n_infers=n_infers, in_channels=12, grid=(24, 24, 8), occupancy=0.12).to(device)
        tk = TeacherKeep(sc, device)
        why = []
        with torch.no_grad():
            x = net.prepare_input(sc.in_feats, sc.in_coords)
            args = (x, sc.global_min_Cs, sc.global_max_Cs, sc.min_Cs, sc.max_Cs)
            got = subnet_parallel_forward(net, *args, keep_override=tk)       # all-gather by subnet: every rank holds all
            _, _, ref = net.ensemble(got, sc.Ts)                              # the single-process ensembler on those logits
            ref_pi = net.panoptic(ref)
            mine = shard_indices(n_infers, rank, world)
            local = dict(got)
            local["panop_predictions"] = [got["panop_predictions"][i] for i in mine]   # what this rank computed itself
            sem, sharded, st = site_sharded_ensemble(net, local, sc.Ts)
            full = gather_sharded(net, sharded)
            if len(full) != len(ref):
                why.append("number of outputs")
            for i, (a, b) in enumerate(zip(full, ref)):                       # BIT for bit
                if not torch.equal(a["voxel_probs"].C, b["voxel_probs"].C):
                    why.append(f"output {i}: rows")
                elif not torch.equal(a["voxel_probs"].F, b["voxel_probs"].F):
                    why.append(f"output {i}: mask probabilities differ by {float((a['voxel_probs'].F - b['voxel_probs'].F).abs().max()):.2e}")
                if not torch.equal(a["sem_probs"].F, b["sem_probs"].F) or not torch.equal(a["query_probs"], b["query_probs"]):
                    why.append(f"output {i}: semantic / query probabilities")
            # the panoptic stage on the sharded rows: areas added over the ranks, label rows gathered
            pis = site_sharded_panoptic(net, sharded)
            for i, (p, r) in enumerate(zip(pis, ref_pi)):
                lab = torch.cat(allgather_rows(p["panoptic"].reshape(-1, 1)))[:, 0]
                semc = torch.cat(allgather_rows(p["semantic"].reshape(-1, 1)))[:, 0]
                info = lambda s: [(v["id"], v["isthing"], v["category_id"], v["query_id"]) for v in s]
                if info(p["segments_infos"]) != info(r["segments_infos"][0]):
                    why.append(f"output {i}: segments {info(p['segments_infos'])} vs {info(r['segments_infos'][0])}")
                if not torch.equal(lab, r["panoptic_seg_sparses"][0]) or not torch.equal(semc, r["semantic_seg_sparses"][0]):
                    why.append(f"output {i}: panoptic / semantic label rows")
            n_seg = sum(len(p["segments_infos"]) for p in pis)
            # bytes: the slab exchange against the all-gather of whole mask tensors
            if world > 1 and not st["bytes_received"] < got["exchange"]["bytes_received"]:
                why.append(f"bytes received {st['bytes_received']} vs all-gather {got['exchange']['bytes_received']}")
        q.put((rank, why, st["bytes_received"], got["exchange"]["bytes_received"], st["rows"], n_seg))
    finally:
        dist.destroy_process_group()


def _run_sharded(world, n_infers, device="cpu"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, n_infers, device)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == world
    for rank, why, sent, gathered, rows, n_seg in res:
        assert not why, f"rank {rank}: {why}"
    print(f"[site-sharded C4] world {world}, M = {n_infers}: bytes RECEIVED per rank {[r[2] for r in res]} (slab exchange) vs "
          f"{[r[3] for r in res]} (all-gather by subnet); rows {res[0][4]}; {res[0][5]} segments")
    assert sum(r[4]["mine"] for r in res) == res[0][4]["union"], "the slabs cover the union rows exactly once"
    return res


def test_site_sharded_ensemble_world2_gloo():
    """The ensembler + panoptic stage of config C4 with the union rows sharded over 2 ranks: bit-identical to the
    single-process stage on the same subnet predictions (masks, semantic rows, query probabilities, segments, labels)."""
    _run_sharded(2, 2)


def test_site_sharded_ensemble_world4_ragged_gloo():
    """M = 6 subnets on 4 ranks: two exchange rounds, the second with two idle senders; 8 slabs over 4 ranks."""
    res = _run_sharded(4, 6)
    assert res[0][5] > 0, "the case should produce panoptic segments"


def test_site_sharded_ensemble_world1_gloo():
    """One rank owns every slab: the sharded code path itself against `Ensembler.ensemble_panop`."""
    _run_sharded(1, 3)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_site_sharded_ensemble_two_ranks_on_one_gpu():
    """Two ranks that SHARE cuda:0 (RCCL refuses two ranks on one device, so the process group is gloo, which stages device
    tensors through the host): the sharded stage end to end on libpascohip.so - resampling, slab exchange, partial sums,
    merge, finish, panoptic kernels with areas added over the ranks - bit-identical to the single-process stage on the GPU."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _run_sharded(2, 4, device="cuda")
