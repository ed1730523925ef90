"""What does the HOST side of a step cost?  The benchmark's step on scenes with ~1 / 30 of the voxels (same grid, same graph, the
same ~1 200 launches, kernels of a few microseconds): the step time is then the Python launch loop + the synchronisations' round
trips, one scene at a time and with three in flight (where the worker threads share the interpreter lock).

    python tools/host_floor.py [occupancy=0.003] > gpurun_out/host_floor.txt
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pasco_amd.graph.serve import SceneServer  # noqa: E402
from pasco_amd.graph.synth import TeacherKeep, make_scene  # noqa: E402


def main():
    occ = float(sys.argv[1]) if len(sys.argv) > 1 else 0.003
    dev = torch.device("cuda", 0)
    net = bench.build_net(3, 283, dev)
    scenes = [make_scene(seed=s, n_infers=3, in_channels=283, occupancy=occ).to(dev) for s in range(4)]
    teachers = [TeacherKeep(sc, dev) for sc in scenes]
    print(f"occupancy {occ}: {[int(s.in_coords[0].shape[0]) for s in scenes]} input points of subnet 0")

    def step(i):
        return bench.run_scene(net, scenes[i % 4], teachers[i % 4])

    import gc
    for k in (1, 3):
        server = SceneServer(dev, step, in_flight=k)
        server.warm(range(4))
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        n = 48
        c0, t0 = time.process_time(), time.perf_counter()
        server.run(range(n), in_flight=k)
        torch.cuda.synchronize()
        dt, dc = time.perf_counter() - t0, time.process_time() - c0
        gc.enable()
        print(f"{k} in flight: {dt / n * 1e3:6.2f} ms per step ({n / dt:5.1f} scenes/s), process CPU {dc / n * 1e3:6.2f} ms per step")
        server.close()


if __name__ == "__main__":
    main()
