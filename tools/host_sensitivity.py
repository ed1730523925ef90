"""How much throughput does a millisecond of HOST time per step cost?  The benchmark's in-flight loop (full-size scenes, three in
flight) with a spin delay of D microseconds added to every convolution launch on the host (110 per step, interpreter lock held):
the slope of the step time over the added host time says what a faster launch path would be worth.

    python tools/host_sensitivity.py [steps=36] > gpurun_out/host_sensitivity.txt
"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pasco_amd.graph.serve import SceneServer  # noqa: E402
from pasco_amd.graph.synth import TeacherKeep, make_scene  # noqa: E402
from pasco_amd.me.backend import hip_backend  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    dev = torch.device("cuda", 0)
    be = hip_backend()
    net = bench.build_net(3, 283, dev)
    scenes = [make_scene(seed=s, n_infers=3, in_channels=283).to(dev) for s in range(4)]
    teachers = [TeacherKeep(sc, dev) for sc in scenes]
    delay = {"us": 0.0, "calls": 0}
    inner = be.conv_fwd

    def conv_fwd(*a, **kw):
        delay["calls"] += 1
        if delay["us"] > 0:
            t_end = time.perf_counter() + delay["us"] * 1e-6
            while time.perf_counter() < t_end:      # holds the interpreter lock, like launch-path Python would
                pass
        return inner(*a, **kw)

    be.conv_fwd = conv_fwd

    def step(i):
        return bench.run_scene(net, scenes[i % 4], teachers[i % 4])

    server = SceneServer(dev, step, in_flight=3)
    server.warm(range(4))
    gc.collect()
    gc.disable()
    for rep in range(2):
        for us in (0.0, 10.0, 20.0, 40.0, 80.0):
            delay["us"], delay["calls"] = us, 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            server.run(range(n), in_flight=3)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            per_step = delay["calls"] / n
            print(f"pass {rep}: +{us:4.0f} us per convolution launch x {per_step:.0f} launches = +{us * per_step * 1e-3:5.2f} ms of host time per step: "
                  f"{dt * 1e3:6.2f} ms per step ({1 / dt:5.1f} scenes/s)", flush=True)
    server.close()


if __name__ == "__main__":
    main()
