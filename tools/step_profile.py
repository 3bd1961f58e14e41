#!/usr/bin/env python3
"""One p_sample step of any configuration, per kernel family and per conv shape (development tool;
run on the GPU box):

    python tools/step_profile.py --batch 1 --res 128 [--image-size 128] [--precision f16x3]
                                 [--steps 20] [--csv gpurun_out/shapes.csv]

Prints the graph-replayed step time (host-timed over `steps` steps) and the HIP-event time per
family with every launch bracketed (SR3 profile mode: no graph, launches serialised by events)."""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
PKG = "3d-super-resolution-face-reconstruction_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--lres", type=int, default=16)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--precision", default="f16f8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--csv", default="")
    a = ap.parse_args()
    if os.environ.get("SR3_LIB"):       # timing experiments with an alternative build of the library
        importlib.import_module(PKG + "._lib").LIB_PATH = os.path.abspath(os.environ["SR3_LIB"])
    synth = importlib.import_module(PKG + ".synth")
    schedule = importlib.import_module(PKG + ".schedule")
    graph = importlib.import_module(PKG + ".graph")
    Engine = importlib.import_module(PKG + ".engine").Engine
    cfg = synth.yml_unet_config(a.image_size)
    e = Engine(cfg, 0)
    e.load_state_dict(synth.synth_state_dict(cfg, 2024))
    e.set_schedule(schedule.schedule_buffers({"schedule": "linear", "n_timestep": a.T, "linear_start": 1e-6, "linear_end": 1e-2}))
    e.set_precision(a.precision)
    B, r = a.batch, a.res
    dc = e.to_device(synth.synth_cond(B, r, a.lres, 3))
    e.sample_begin(dc.ptr, B, r, r, None, 7, 0)
    t = a.T - 1
    for _ in range(4):
        e.sample_step(t); t -= 1
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        e.sample_step(t); t -= 1
    e.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    gf = graph.flops_per_image(cfg, r, r) * B / 1e9
    print(f"B={B} {r}x{r} image_size={a.image_size} [{a.precision}]: {dt * 1e3:.3f} ms/step (graph replay) "
          f"= {gf / dt / 1e3:.1f} TFLOP/s algorithmic; {B / (a.T * dt):.3f} img/s at T={a.T}")
    e.profile_reset(); e.profile_enable(True)
    for _ in range(min(a.steps, 10)):
        e.sample_step(t); t -= 1
    prof = e.profile_get()
    n = min(a.steps, 10)
    tot = sum(v["ms"] for v in prof.values()) / n
    print(f"  event-timed (no graph): {tot:.3f} ms/step; " +
          "; ".join(f"{k} {v['ms'] / n:.3f} ms ({v['launches'] // n} launches)" for k, v in prof.items()))
    if a.csv:
        e.profile_dump_csv(a.csv)
    e.profile_enable(False)
    e.close()


if __name__ == "__main__":
    main()
