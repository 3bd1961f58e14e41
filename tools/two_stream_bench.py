#!/usr/bin/env python3
"""Experiment: does splitting the batch over two contexts / streams (so that one half's HBM-bound
GroupNorm passes overlap the other half's MFMA-bound convolutions) raise throughput?"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
PKG = "3d-super-resolution-face-reconstruction_amd"
synth = importlib.import_module(PKG + ".synth")
schedule = importlib.import_module(PKG + ".schedule")
Engine = importlib.import_module(PKG + ".engine").Engine
cfg = synth.yml_unet_config(224)
sd = synth.synth_state_dict(cfg, 7)
sched = schedule.schedule_buffers({"schedule": "linear", "n_timestep": 1000, "linear_start": 1e-6, "linear_end": 1e-2})
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"

def mk(B):
    e = Engine(cfg, 0); e.load_state_dict(sd); e.set_schedule(sched); e.set_precision(prec)
    c = e.to_device(synth.synth_cond(B, 128, 16, 1))
    e.sample_begin(c.ptr, B, 128, 128, None, 1, 0)
    e._cond = c
    return e

def run(engs, steps):
    for e in engs:
        for t in range(3): e.sample_step(999 - t)
    for e in engs: e.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for e in engs: e.sample_step(990 - k)
    for e in engs: e.synchronize()
    return (time.perf_counter() - t0) / steps

one = [mk(64)]
t1 = run(one, 20)
print(f"[{prec}] 1 ctx  x B=64: {t1*1e3:.2f} ms per 64-image step -> {64/(1000*t1):.3f} img/s")
one[0].close()
two = [mk(32), mk(32)]
t2 = run(two, 20)
print(f"[{prec}] 2 ctx  x B=32: {t2*1e3:.2f} ms per 64-image step -> {64/(1000*t2):.3f} img/s")
for e in two: e.close()
three = [mk(22), mk(21), mk(21)]
t3 = run(three, 20)
print(f"[{prec}] 3 ctx  x B~21: {t3*1e3:.2f} ms per 64-image step -> {64/(1000*t3):.3f} img/s")
