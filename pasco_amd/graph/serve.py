"""Several scenes in flight on one GPU, and the memory they run in.

A step of the hot path has ~20 host synchronisations (row counts of map events, the Hungarian step); one stream alone
leaves the GPU idle during them.  `SceneServer` runs `in_flight` worker threads, each bound to its own HIP stream, over
ONE net (shared weights and operand caches; per-stream workspaces, status words and query-side hipGraphs live in the
backend / the modules, keyed by stream).  Reference counterpart: the Lightning predict loop around
`Net.step_inference` (pasco/models/net_panoptic_sparse.py:539-576), which serves one scene at a time.

Memory: every buffer of a step comes from torch's caching allocator, per stream.  A device allocation inside a served loop
is what rounds 2 - 3 called the "slow mode": a request the cache cannot serve goes to the driver, which costs ~0.02 ms on
some boxes of the pool and several ms on others, stalls the launch that waits for it and - several scenes in flight - every
stream; it hit whichever launches allocated the step's largest outputs (the 970 MB K / V operands then; profiles/README.md,
round 4).  `warm` therefore runs every scene shape on every stream (graph captures one at a time) and then the in-flight
loop itself until a whole round needed no device allocation, with requests rounded to 1/8 of a power of two so that scenes
of slightly different size reuse each other's blocks.  `vet_cached_blocks` is the diagnostic that ruled the other suspect
out (slow physical memory behind some blocks): it write-tests every large block the allocator holds after the warm-up,
quarantines and replaces a block that streams at less than half the rate of its peers, and reports the rates.
"""
from __future__ import annotations

import sys
import threading
import time
from typing import Callable, List, Optional, Sequence

import torch

_QUARANTINE: List[torch.Tensor] = []     # blocks that failed the write test: held for the life of the process


def _device_mallocs(device) -> int:
    try:
        return int(torch.cuda.memory_stats(device).get("num_device_alloc", 0))
    except Exception:
        return -1


def _fill_rate(t: torch.Tensor, stream: torch.cuda.Stream, reps: int = 2) -> float:
    """GB/s of a plain streaming write over `t` on `stream` (best of `reps`)."""
    best = 0.0
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        t.fill_(0)
        e1.record(stream)
        e1.synchronize()
        best = max(best, t.numel() * t.element_size() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def vet_cached_blocks(device, streams: Sequence[torch.cuda.Stream], min_bytes: int = 128 << 20, floor_frac: float = 0.5,
                      max_replace: int = 3) -> dict:
    """Write-test every inactive block of at least `min_bytes` in the caching allocator's pools of `streams` (after a
    warm-up: those are the blocks the next steps' large tensors will live in).  Blocks slower than `floor_frac` x the median
    rate are quarantined and replaced (a replacement is a fresh device allocation, tested the same way, at most `max_replace`
    times per block).  Returns the rates and what was done; costs one pass over the cached memory (~10 ms per 30 GB)."""
    device = torch.device(device)
    torch.cuda.synchronize(device)
    by_stream = {}
    for seg in torch.cuda.memory_snapshot():
        if seg.get("device") != device.index:
            continue
        for b in seg.get("blocks", ()):
            if b.get("state") == "inactive" and b.get("size", 0) >= min_bytes:
                by_stream.setdefault(int(seg.get("stream", 0)), []).append(int(b["size"]))
    report = {"min_block_MB": min_bytes >> 20, "blocks": 0, "GB": 0.0, "quarantined": 0, "replaced": 0, "rates_GBps": []}
    held, rates = [], []
    for st in streams:
        sizes = sorted(by_stream.get(int(st.cuda_stream), ()), reverse=True)
        with torch.cuda.stream(st):
            for sz in sizes:      # hold every block while testing so that each request lands in a different one
                t = torch.empty(sz, dtype=torch.uint8, device=device)
                held.append((st, t))
                rates.append(_fill_rate(t, st))
    if not rates:
        return report
    srt = sorted(rates)
    median = srt[len(srt) // 2]
    report.update(blocks=len(rates), GB=round(sum(t.numel() for _, t in held) / 2 ** 30, 2),
                  rates_GBps=[round(srt[0], 1), round(median, 1), round(srt[-1], 1)])
    for i, ((st, t), r) in enumerate(zip(list(held), rates)):
        tries = 0
        while r < floor_frac * median and tries < max_replace:
            _QUARANTINE.append(t)
            report["quarantined"] += 1
            with torch.cuda.stream(st):
                t = torch.empty(t.numel(), dtype=torch.uint8, device=device)      # nothing of that size is free: a new allocation
                r = _fill_rate(t, st)
            held[i] = (st, t)
            report["replaced"] += 1
            tries += 1
        if r < floor_frac * median:
            report.setdefault("still_slow_GBps", []).append(round(r, 1))
    del held, t
    torch.cuda.synchronize(device)
    return report


class SceneServer:
    """`in_flight` scenes at a time through `step(item)` on one GPU.  `step` is called under `torch.no_grad()` with the
    worker's stream current; items are whatever `step` takes (the benchmark passes scene indices)."""

    def __init__(self, device, step: Callable, in_flight: int = 3, switch_interval_ms: Optional[float] = 0.5,
                 allocator_rounding: Optional[int] = 8):
        self.device = torch.device(device)
        # Request sizes differ from scene to scene by a few percent (row counts), so a cached block rarely fits the next
        # scene's request exactly; rounding requests up to 1/8 of a power of two lets scenes reuse each other's blocks and
        # the pools settle in one pass instead of growing for many.  PROCESS-WIDE allocator setting (up to 12.5 % more memory
        # per block for every model of the process; it cannot be read back, so `close` does not undo it): pass
        # allocator_rounding=None to leave the allocator alone ("Serving loop")
        self.allocator_rounding = None
        if allocator_rounding:
            try:
                torch.cuda.memory._set_allocator_settings(f"roundup_power2_divisions:{int(allocator_rounding)}")
                self.allocator_rounding = int(allocator_rounding)
            except Exception:      # an optimisation of the warm-up, never a requirement
                pass
        self.step = step
        self.in_flight = max(int(in_flight), 1)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.in_flight)]
        self._switch_interval_before = None
        if self.in_flight > 1 and switch_interval_ms:
            # worker threads hand the interpreter over every 0.5 ms instead of every 5 ms: a scene whose stream has run dry
            # gets to enqueue sooner (process-wide interpreter setting: `close` puts the previous value back)
            self._switch_interval_before = sys.getswitchinterval()
            sys.setswitchinterval(switch_interval_ms * 1e-3)
        for s in self.streams:                      # whatever was uploaded on the current stream is visible to the workers
            s.wait_stream(torch.cuda.current_stream(self.device))

    def close(self) -> None:
        """Retire the worker streams: their per-stream buffers in the backend (workspaces, status pairs) are dropped, so a
        long-lived process that builds servers repeatedly neither leaks them nor hands a recycled stream handle stale flags."""
        from ..me.backend import backend_for
        try:
            be = backend_for(self.device)
        except Exception:
            return
        from .transformer import release_stream_graphs
        for s in self.streams:
            s.synchronize()
            be.release_stream(s)
            release_stream_graphs(s)       # the query-side hipGraphs captured for this stream
        self.streams = []
        if self._switch_interval_before is not None:
            sys.setswitchinterval(self._switch_interval_before)
            self._switch_interval_before = None

    # -- one at a time, on the caller's stream -------------------------------------------------------------------------
    def run_serial(self, items: Sequence, on_done: Optional[Callable] = None):
        last = None
        with torch.no_grad():
            for it in items:
                last = self.step(it)
                if on_done is not None:
                    on_done(it, last)
        return last

    # -- in flight -----------------------------------------------------------------------------------------------------
    def run(self, items: Sequence, in_flight: Optional[int] = None, on_done: Optional[Callable] = None):
        """Every item once; worker w takes the next item whenever it is free.  Returns the result of the item that finished
        last, after every worker's stream has drained.  `on_done(item, result)` runs under the server's lock."""
        k = self.in_flight if in_flight is None else max(1, min(int(in_flight), self.in_flight))
        if k <= 1:
            return self.run_serial(items, on_done)
        items = list(items)
        lock = threading.Lock()
        state = {"next": 0, "last": None}
        errors: List[BaseException] = []

        def worker(w: int):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[w]), torch.no_grad():
                    while True:
                        with lock:
                            i = state["next"]
                            state["next"] += 1
                        if i >= len(items) or errors:
                            break
                        out = self.step(items[i])
                        with lock:
                            state["last"] = out
                            if on_done is not None:
                                on_done(items[i], out)
                self.streams[w].synchronize()
            except BaseException as e:      # surfaced in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(k)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return state["last"]

    # -- warm-up -------------------------------------------------------------------------------------------------------
    def warm(self, items: Sequence, max_rounds: int = 8, quiet_rounds: int = 2) -> dict:
        """Every item on the caller's stream until a whole pass needed no device allocation (map shapes, operand caches, the
        stream's allocator pool), once on every worker stream (their graphs and workspaces; one stream at a time - graph
        captures do not overlap other launches), then rounds of the in-flight loop over 2 x in_flight x len(items) steps
        until `quiet_rounds` rounds in a row needed no device allocation either (the pools depend on the order the workers draw
        the items in; they settle within a few rounds - one quiet round still left 0 - 5 allocations to a following loop of 24
        steps, `profiles/r4x_*`).  Why it matters: a device allocation inside a served loop costs from ~0.1 ms
        to several ms depending on the box, stalls the launch that waits for it and, on the slow boxes, every stream
        (profiles/README.md, round 4: the "slow mode" of rounds 2 - 3)."""
        items = list(items)
        t0 = time.perf_counter()
        serial_rounds, quiet1 = 0, False
        while not quiet1 and serial_rounds < max_rounds:
            before = _device_mallocs(self.device)
            self.run_serial(items)
            serial_rounds += 1
            quiet1 = serial_rounds > 1 and _device_mallocs(self.device) == before
        if self.in_flight > 1:
            for s in self.streams:
                with torch.cuda.stream(s):
                    self.run_serial(items)
                s.synchronize()
        rounds, streak = 0, 0
        quiet = self.in_flight <= 1
        while not quiet and rounds < max_rounds:
            before = _device_mallocs(self.device)
            self.run([items[i % len(items)] for i in range(2 * self.in_flight * len(items))])
            rounds += 1
            streak = streak + 1 if _device_mallocs(self.device) == before else 0
            quiet = streak >= max(int(quiet_rounds), 1)
        torch.cuda.synchronize(self.device)
        return {"serial_rounds": serial_rounds, "serial_settled": bool(quiet1), "in_flight_rounds": rounds,
                "settled": bool(quiet), "seconds": round(time.perf_counter() - t0, 2)}
