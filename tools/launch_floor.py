#!/usr/bin/env python3
"""What a chain of N dependent kernels costs on this box when the kernels do (almost) nothing: the launch-latency floor of
one B = 1 sampler step (~125 dependent launches, profiles/README.md finding 69). hipGraph replay of N one-element torch
kernels on one stream (every node depends on the one before, like the layers of a UNet at batch 1), N = 1 .. 250.

    python tools/launch_floor.py            (GPU box)
"""
import time

import torch


def chain_ms(n, reps=200):
    x = torch.zeros(64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            for _ in range(n):
                x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    prev = None
    for n in (1, 25, 50, 125, 250):
        ms = chain_ms(n)
        print(f"graph of {n:4d} dependent one-wave kernels: {ms:.4f} ms per replay = {ms / n * 1e3:.2f} us per kernel")
