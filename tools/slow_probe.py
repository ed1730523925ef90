"""Which memory is slow on this box?  (The "slow mode" of rounds 2 - 3: whole runs in which only the launches WRITING the
step's largest buffers - the 970 MB K / V operands of the finest level - were ~7 x slow.)

Allocates `blocks` x 1 GiB through torch's caching allocator (all held at once, so every block is its own device
allocation), and for each block times
  fill   a plain streaming write (torch fill kernel)
  k1     the launch that was slow: the 64 -> 384 projection of 631 626 rows on k_conv_dma writing ONLY its split f16 operand
         (970 MB) into the block
  read   a streaming read of the block (sum)
then the same through ONE 8 GiB allocation cut into 1 GiB slices (a different size class for the driver), and prints GB/s
per block.  Rates well below the others' = the block's physical memory, not the kernels.

    python tools/slow_probe.py [blocks=12] > gpurun_out/slow_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pasco_amd.me.backend import ACT_NONE, hip_backend   # noqa: E402

GIB = 1 << 30


def timed(fn, reps=3):
    best = float("inf")
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    be = hip_backend()
    g = torch.Generator().manual_seed(0)
    n, cin, cout = 631626, 64, 384
    x = torch.randn(n, cin, generator=g).to(dev)
    w = (torch.randn(cin, cout, generator=g) * 0.05).to(dev)
    bias = torch.zeros(cout, device=dev)
    x_split = be.split_rows(x)
    w_split = be.split_weight_rows(w)
    out_bytes = n * cout * 4                      # hi + lo f16 per channel
    free0, total = torch.cuda.mem_get_info(dev)
    print(f"device memory: {free0 / GIB:.1f} GiB free of {total / GIB:.1f}; k1 launch writes {out_bytes / 1e6:.0f} MB per block")

    def probe(name, tensors):
        rows = []
        for i, t in enumerate(tensors):
            osp = t[:out_bytes].view(torch.float16).view(n, cout // 32, 2, 32)
            t_fill = timed(lambda: t.fill_(0))
            t_k1 = timed(lambda: be.conv_fwd(None, w, None, n, xshape=(n, cin), bias=bias, split=w_split, in_split=x_split,
                                             emit_split=(None, None, ACT_NONE), want_out=False, out_split=osp))
            t_read = timed(lambda: t.view(torch.int32).sum())
            rows.append((t.data_ptr(), GIB / t_fill / 1e9, (out_bytes + n * cin * 4) / t_k1 / 1e9, t_k1 * 1e6, GIB / t_read / 1e9))
        fills = sorted(r[1] for r in rows)
        k1s = sorted(r[2] for r in rows)
        print(f"== {name}: {len(rows)} blocks; fill median {fills[len(fills) // 2]:.0f} GB/s (min {fills[0]:.0f}), "
              f"k1 median {k1s[len(k1s) // 2]:.0f} GB/s (min {k1s[0]:.0f})")
        for ptr, f, k, us, r in rows:
            flag = "  <-- SLOW" if (f < 0.5 * fills[len(fills) // 2] or k < 0.5 * k1s[len(k1s) // 2]) else ""
            print(f"  {ptr:#016x}  fill {f:7.0f} GB/s   k1 {k:7.0f} GB/s ({us:7.0f} us)   read {r:7.0f} GB/s{flag}")

    # what a device allocation costs on this box (host time of a request the cache cannot serve, and of the first launch
    # into the new block): the cost a served loop pays for every allocation its warm-up did not provoke
    import time
    for mb in (64, 256, 1024):
        ts, tl = [], []
        for _ in range(5):
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
            t1 = time.perf_counter()
            t[:1 << 20].fill_(0)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0) * 1e3)
            tl.append((t2 - t1) * 1e3)
            del t
        t0 = time.perf_counter()
        torch.cuda.empty_cache()
        tf = (time.perf_counter() - t0) * 1e3
        print(f"device allocation of {mb:5d} MB: {min(ts):7.3f} .. {max(ts):7.3f} ms (host), first 1 MB fill + sync "
              f"{min(tl):6.3f} .. {max(tl):6.3f} ms, free {tf:6.3f} ms")
    held = [torch.empty(GIB, dtype=torch.uint8, device=dev) for _ in range(blocks)]
    probe("caching allocator, 1 GiB requests", held)
    st = torch.cuda.memory_stats(dev)
    print(f"device mallocs so far: {st.get('num_device_alloc', 0)}, reserved {st.get('reserved_bytes.all.current', 0) / GIB:.1f} GiB")
    del held
    torch.cuda.empty_cache()
    big = torch.empty(8 * GIB, dtype=torch.uint8, device=dev)
    probe("one 8 GiB allocation, 1 GiB slices", [big[i * GIB:(i + 1) * GIB] for i in range(8)])
    del big
    torch.cuda.empty_cache()
    # the same 1 GiB requests again after the cache was returned to the driver: does a fresh allocation land elsewhere?
    held = [torch.empty(GIB, dtype=torch.uint8, device=dev) for _ in range(min(blocks, 6))]
    probe("caching allocator again (after empty_cache)", held)


if __name__ == "__main__":
    main()
