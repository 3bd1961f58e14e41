"""Batch / resolution sweep of the sampler (yml-literal UNet): images of a batched run equal the same
images sampled alone (Philox keyed by the global image index) to fp32 round-off. Different batch sizes
take different tile shapes, kernels (x-halo / generic, split-K), GroupNorm statistics paths and
split-only decisions, so this pins them against each other; the oracle comparisons elsewhere pin the
small-batch paths to the reference."""
import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")
schedule = pkg("schedule")


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_batch_resolution_sweep(prec):
    cfg = synth.yml_unet_config(224)
    e = pkg("engine").Engine(cfg, 0)
    e.load_state_dict(synth.synth_state_dict(cfg, 3))
    e.set_schedule(schedule.schedule_buffers({"schedule": "linear", "n_timestep": 3, "linear_start": 1e-4,
                                              "linear_end": 2e-2}))
    e.set_precision(prec)
    worst = 0.0
    for r in (128, 32):
        for B in (3, 16, 40):
            cond = synth.synth_cond(B, r, max(4, r // 8), 5)
            full = e.sample_np(cond, seed=11)
            for i in (0, B - 1):
                alone = e.sample_np(cond[i:i + 1], seed=11, image_offset=i)
                d = float(np.abs(alone[0] - full[i]).max())
                assert d <= 2e-5, (prec, r, B, i, d)
                worst = max(worst, d)
    e.close()
    print(f"[{prec}] worst max-abs difference {worst:.2e}")
