#!/usr/bin/env python3
"""Does running the batch as two half-batches on two streams overlap one half's HBM-bound GroupNorm passes with the other
half's MFMA-bound convs? Two Engine contexts (32 images each, own stream, own hipGraph) stepped alternately from one host
thread, against one context with 64 images. Development tool (profiles/README.md finding 62); run on the GPU box."""
import importlib, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
PKG = "3d-super-resolution-face-reconstruction_amd"
import torch
synth = importlib.import_module(PKG + ".synth")
schedule = importlib.import_module(PKG + ".schedule")
Engine = importlib.import_module(PKG + ".engine").Engine

T, r, K, W = 1000, 128, 30, 5
cfg = synth.yml_unet_config(224)
sd = synth.synth_state_dict(cfg, 2024)
sched = schedule.schedule_buffers({"schedule": "linear", "n_timestep": T, "linear_start": 1e-6, "linear_end": 1e-2})


def make(B, stream):
    e = Engine(cfg, 0)
    e.load_state_dict(sd)
    e.set_schedule(sched)
    e.set_precision("f16x3")
    e.set_stream(stream.cuda_stream)
    cond = torch.from_numpy(synth.synth_cond(B, r, 16, 1)).cuda()
    e.sample_begin(cond.data_ptr(), B, r, r, None, seed=7, image_offset=0)
    return e, cond


def run(engs, n):
    t = T - 1
    for _ in range(n):
        for e, _c in engs:
            e.sample_step(t, None)
        t -= 1


for label, parts in (("1 x 64", [64]), ("2 x 32", [32, 32]), ("4 x 16", [16, 16, 16, 16])):
    streams = [torch.cuda.Stream() for _ in parts]
    engs = [make(b, s) for b, s in zip(parts, streams)]
    run(engs, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(engs, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{label}: {dt * 1e3:.3f} ms per step of 64 images = {64 / (T * dt):.3f} img/s")
    for e, _c in engs:
        e.close()
