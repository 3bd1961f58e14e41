#!/usr/bin/env python3
"""Development check (GPU box): batch / resolution sweep of the sampler. For every (B, r) the first and
last image of a batched 3-step run must equal the same images sampled alone (B = 1, Philox keyed by
the global image index) to fp32 round-off — different batch sizes take different tile shapes, kernels
(x-halo / generic, split-K), GroupNorm statistics paths and split-only decisions."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
PKG = "3d-super-resolution-face-reconstruction_amd"
synth = importlib.import_module(PKG + ".synth"); schedule = importlib.import_module(PKG + ".schedule")
Engine = importlib.import_module(PKG + ".engine").Engine

cfg = synth.yml_unet_config(int(sys.argv[1]) if len(sys.argv) > 1 else 224)
e = Engine(cfg, 0); e.load_state_dict(synth.synth_state_dict(cfg, 3))
e.set_schedule(schedule.schedule_buffers({"schedule": "linear", "n_timestep": 3, "linear_start": 1e-4, "linear_end": 2e-2}))
worst = 0.0
for prec in ("f16x3", "f32"):
    e.set_precision(prec)
    for r in (128, 64, 32):
        for B in (3, 8, 16, 40, 64):
            cond = synth.synth_cond(B, r, max(4, r // 8), 5)
            full = e.sample_np(cond, seed=11)
            for i in (0, B - 1):
                alone = e.sample_np(cond[i:i + 1], seed=11, image_offset=i)
                d = float(np.abs(alone[0] - full[i]).max())
                worst = max(worst, d)
                if d > 2e-5:
                    print(f"MISMATCH prec={prec} r={r} B={B} image {i}: {d:.3e}")
    print(f"[{prec}] done, worst so far {worst:.3e}", flush=True)
print("worst max-abs difference:", worst)
sys.exit(0 if worst <= 2e-5 else 1)
