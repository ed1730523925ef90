"""Host-side cost of one step: cProfile over K sequential steps (GPU box).  Prints the top functions by own time and by
cumulative time; the wall per step beside the sum of host time says how far the host is from being the bottleneck."""
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pasco_amd.graph.synth import make_scene, TeacherKeep  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    occ = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10      # 0.003: kernels of a few us - the wall IS the host side
    dev = torch.device("cuda", 0)
    net = bench.build_net(3, 283, dev)
    scenes = [make_scene(seed=s, n_infers=3, in_channels=283, occupancy=occ).to(dev) for s in range(2)]
    teachers = [TeacherKeep(sc, dev) for sc in scenes]
    with torch.no_grad():
        for i in range(4):
            bench.run_scene(net, scenes[i % 2], teachers[i % 2])
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        for i in range(k):
            bench.run_scene(net, scenes[i % 2], teachers[i % 2])
        pr.disable()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
    print(f"wall per step under cProfile: {dt * 1e3:.1f} ms")
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
        print(s.getvalue())


if __name__ == "__main__":
    main()
