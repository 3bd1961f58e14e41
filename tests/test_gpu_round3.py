"""Round-3 parity / behaviour cases (all through the C-ABI):

* BASELINE config 2 — the benchmarked workload — pinned to the REFERENCE over its full horizon: 16 -> 128,
  yml-literal UNet, T = 1000, B = 1 (tests/golden/sampler_cfg2_16_128_T1000.npz, a run of
  model/sr/sr3_modules/diffusion.py:189-215 itself), both arithmetic modes, bar 1e-3;
* the split-K reduce pass on sizes whose blocks straddle two images (14x14, 20x20: the deepest levels of a
  224 / 160-pixel run) — ADVICE round 2;
* the split-f16 range limit no longer fails the call: default policy finishes it in exact f32
  (the reference is plain fp32, unet.py:235-265), strict policy keeps the failure;
* the RCCL branch of the product's collective executed once, with one rank (librccl initialisation on this pool);
* the torch facade orders its work against torch's stream with events (no host synchronisation).
"""
import os
import socket
import warnings

import numpy as np
import pytest

import sr3_oracle as oracle
from conftest import cfg_from_meta, load_golden, pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")
schedule = pkg("schedule")
metrics = pkg("metrics")
graph = pkg("graph")
_lib = pkg("_lib")
Sr3Error, Sr3RangeWarning = _lib.Sr3Error, _lib.Sr3RangeWarning
BAR = 1e-3
PRECISIONS = ["f32", "f16x3"]


def _engine(cfg, sd, prec, sched_opt=None):
    e = pkg("engine").Engine(cfg, 0)
    e.load_state_dict(sd)
    e.set_precision(prec)
    if sched_opt:
        e.set_schedule(schedule.schedule_buffers(sched_opt))
    return e


@pytest.mark.parametrize("prec", PRECISIONS)
def test_sampler_golden_cfg2_T1000_full_horizon(prec):
    """The headline configuration against the reference's own 1000-step run (injected noise)."""
    g = load_golden("sampler_cfg2_16_128_T1000.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    B, r, T, st = m["B"], m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    assert (B, r, T, m["l"]) == (1, 128, 1000, 16)
    e = _engine(cfg, synth.synth_state_dict(cfg, m["seed"]), prec, m["schedule"])
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    final, frames = e.sample_np(g["cond"], noise=noise, frames=True)
    assert frames.shape == (10, B, 3, r, r)
    err = np.abs(frames[..., ::st, ::st] - g["frames_sub"]).reshape(10, -1).max(1)
    e_fin = np.abs(final - g["final"]).max()
    ps = [metrics.batch_psnr_stats(frames[f][..., ::st, ::st], g["frames_sub"][f]) for f in range(10)]
    print(f"cfg2 16->128 T=1000 [{prec}]: per-frame max abs err (sub-sampled) {np.array2string(err, precision=2)}; "
          f"final (every pixel) {e_fin:.2e}; PSNR of the final image {metrics.batch_psnr_stats(final, g['final'])}; "
          f"per-frame PSNR (sub-sampled) {[p['mean_db'] for p in ps]} identical {[p['identical'] for p in ps]}")
    assert err.max() <= BAR and e_fin <= BAR
    np.testing.assert_array_equal(final, frames[-1])
    assert e.fallback_calls() == 0
    e.close()


STRADDLE_CASES = [
    # B, H, W, Cin, Cout: two-kernel split-K whose reduce blocks (TP = HW / 64 pixels) straddle two images
    (2, 14, 14, 512, 512),
    (3, 20, 20, 256, 256),
    (2, 28, 28, 256, 256),
]


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("case", STRADDLE_CASES)
def test_splitk_reduce_blocks_straddling_images(case, prec):
    B, H, W, Cin, Cout = case
    e = pkg("engine").Engine(synth.tiny_unet_config(), 0)
    e.set_precision(prec)
    rs = np.random.RandomState(H * 131 + Cin)
    x = rs.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    cb = rs.standard_normal((B, Cout)).astype(np.float32)        # FeatureWiseAffine bias: differs per image
    res = rs.standard_normal((B, H, W, Cout)).astype(np.float32)
    got = e.op_conv2d(x, w, b, chan_bias=cb, resid=res)
    want = oracle.conv2d(x, w, b) + cb[:, None, None, :] + res
    err = np.abs(got - want).max()
    assert err <= 2e-5 * max(1.0, np.abs(want).max()), err
    e.close()


def _overflow_net():
    cfg = synth.tiny_unet_config()
    sd = synth.synth_state_dict(cfg, 77)
    sd["downs.0.weight"] = sd["downs.0.weight"] * np.float32(3e5)       # first conv output ~1e5..1e6: beyond fp16
    return cfg, sd


def test_overflow_is_finished_in_f32_by_default_and_fails_when_strict():
    cfg, sd = _overflow_net()
    rs = np.random.RandomState(5)
    x = rs.standard_normal((2, 6, 16, 16)).astype(np.float32)
    nl = np.array([0.3, 0.7], np.float32)
    want = oracle.unet_forward(sd, cfg, x, nl)
    e = _engine(cfg, sd, "f16x3")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = e.unet_forward_np(x, nl)
    assert any(issubclass(w.category, Sr3RangeWarning) and "exact f32" in str(w.message) for w in rec)
    assert e.fallback_calls() == 1
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    # an in-range input on the same context: no fallback, still the split-f16 mode
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        small = e.unet_forward_np(x * np.float32(1e-9), nl)
    assert np.isfinite(small).all() and e.fallback_calls() == 1
    # whole sampler call: checkpointed replay in f32 == an all-f32 run of the same call
    sched = {"schedule": "linear", "n_timestep": 24, "linear_start": 1e-4, "linear_end": 2e-2}
    e.set_schedule(schedule.schedule_buffers(sched))
    cond, noise = synth.synth_cond(2, 16, 8, 1), synth.synth_noise(24, 2, 3, 16, 16, 1)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fin, frames = e.sample_np(cond, noise=noise, frames=True)
    assert any(issubclass(w.category, Sr3RangeWarning) for w in rec) and e.fallback_calls() == 2
    e.set_precision("f32")
    fin32, frames32 = e.sample_np(cond, noise=noise, frames=True)
    np.testing.assert_array_equal(fin, fin32)
    np.testing.assert_array_equal(frames, frames32)
    # Philox noise replays exactly too (draws are keyed by the step index)
    e.set_precision("f16x3")
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        p16 = e.sample_np(cond, seed=99)
    e.set_precision("f32")
    np.testing.assert_array_equal(p16, e.sample_np(cond, seed=99))
    # strict policy: the round-2 behaviour
    e.set_precision("f16x3")
    e.set_range_policy(True)
    with pytest.raises(Sr3Error, match="fp16 range"):
        e.unet_forward_np(x, nl)
    with pytest.raises(Sr3Error, match="fp16 range"):
        e.sample_np(cond, noise=noise)
    e.close()


def test_overflow_late_in_the_loop_replays_from_the_last_checkpoint():
    """The range flag trips in a LATER segment only: the net is in range for ordinary states, and one injected
    noise slab late in the loop is scaled up so that x_t (and with it the first conv's output) leaves the fp16
    range from there on. The segments before it stay those of the split-f16 run, the rest is the f32 replay from
    the last clean checkpoint: the whole result is within the parity bar of an all-f32 run of the same call."""
    cfg, sd = _overflow_net()
    sd["downs.0.weight"] = sd["downs.0.weight"] / np.float32(3e5) * np.float32(8e3)    # in range for |x_t| <~ 8
    T = 40
    sched = {"schedule": "linear", "n_timestep": T, "linear_start": 1e-4, "linear_end": 2e-2}
    e = _engine(cfg, sd, "f16x3", sched)
    cond, noise = synth.synth_cond(2, 16, 8, 3), synth.synth_noise(T, 2, 3, 16, 16, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # ordinary noise: no fallback
        e.sample_np(cond, noise=noise)
    noise[26] *= np.float32(300.0)                          # the draw of step t = T - 26 = 14: x_13 becomes huge
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fin, frames = e.sample_np(cond, noise=noise, frames=True)
    assert any(issubclass(w.category, Sr3RangeWarning) for w in rec) and e.fallback_calls() == 1
    e.set_precision("f32")
    fin32, frames32 = e.sample_np(cond, noise=noise, frames=True)
    d = np.abs(frames - frames32).reshape(frames.shape[0], -1).max(1)
    print(f"late overflow: per-frame max |f16x3 + f32 replay - all f32| = {np.array2string(d, precision=2)}")
    assert np.isfinite(fin).all() and np.abs(fin - fin32).max() <= 1e-4 and d.max() <= 1e-4 * max(1.0, np.abs(frames32).max())
    e.close()


def test_packed_state_overflow_from_injected_noise_is_caught():
    """the DDPM update writes the packed split-f16 copy of x_t; an injected noise slab large enough to push
    x_t itself beyond the fp16 range must raise the flag there (not surface as inf / NaN images): the call is finished
    in f32 and equals the all-f32 run."""
    cfg = synth.tiny_unet_config()
    sd = synth.synth_state_dict(cfg, 12)
    T = 12
    sched = {"schedule": "linear", "n_timestep": T, "linear_start": 1e-4, "linear_end": 2e-2}
    e = _engine(cfg, sd, "f16x3", sched)
    cond, noise = synth.synth_cond(2, 16, 8, 4), synth.synth_noise(T, 2, 3, 16, 16, 4)
    noise[5] *= np.float32(3e6)                      # sigma ~ 0.1: x_t ~ 1e5..1e6 after that step
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        fin = e.sample_np(cond, noise=noise)
    assert any(issubclass(w.category, Sr3RangeWarning) for w in rec)
    e.set_precision("f32")
    fin32 = e.sample_np(cond, noise=noise)
    assert np.isfinite(fin).all()
    np.testing.assert_allclose(fin, fin32, atol=1e-4, rtol=0)
    e.close()


def test_final_conv_on_the_fly_split_is_range_checked():
    """final_conv builds its split-f16 operands on the fly (GroupNorm + Swish inside the kernel); a value beyond
    the fp16 range there must raise the flag like every stored one. A huge final GroupNorm gamma makes swish(a x + b)
    ~1e6: the default policy finishes the forward in f32 (== oracle), the strict policy fails."""
    cfg = synth.tiny_unet_config()
    sd = synth.synth_state_dict(cfg, 21)
    sd["final_conv.block.0.weight"] = sd["final_conv.block.0.weight"] * np.float32(3e5)
    sd["final_conv.block.3.weight"] = sd["final_conv.block.3.weight"] * np.float32(1e-5)     # keep eps O(1)
    rs = np.random.RandomState(6)
    x = rs.standard_normal((2, 6, 16, 16)).astype(np.float32)
    nl = np.array([0.4, 0.6], np.float32)
    want = oracle.unet_forward(sd, cfg, x, nl)
    e = _engine(cfg, sd, "f16x3")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = e.unet_forward_np(x, nl)
    assert any(issubclass(w.category, Sr3RangeWarning) for w in rec), "the on-the-fly split did not raise the range flag"
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    e.set_range_policy(True)
    with pytest.raises(Sr3Error, match="fp16 range"):
        e.unet_forward_np(x, nl)
    e.close()


def test_facade_finishes_overflowing_calls():
    """define_G facade, default policy: super_resolution / denoise_fn / p_sample finish like the reference."""
    import torch
    P = pkg()
    cfg, sd = _overflow_net()
    sched = {"schedule": "linear", "n_timestep": 8, "linear_start": 1e-4, "linear_end": 2e-2}
    opt = {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": cfg.inner_channel,
                 "channel_multiplier": list(cfg.channel_mults), "attn_res": list(cfg.attn_res),
                 "res_blocks": cfg.res_blocks, "dropout": 0.0},
        "beta_schedule": {"train": sched, "val": sched},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": True}}}}
    netG = P.define_G(opt).cuda()
    netG.load_state_dict({"denoise_fn." + k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    netG.set_new_noise_schedule(sched, [0])
    assert netG.denoise_fn.precision == "f16f8"
    cond, noise = synth.synth_cond(2, 16, 8, 2), synth.synth_noise(8, 2, 3, 16, 16, 2)
    want, _ = oracle.p_sample_loop(sd, cfg, oracle.noise_schedule(sched), cond, noise)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = netG.super_resolution_batch(torch.from_numpy(cond).cuda(), noise=torch.from_numpy(noise))
        x1 = netG.p_sample(torch.from_numpy(noise[0]).cuda(), 7, condition_x=torch.from_numpy(cond).cuda(),
                           noise=torch.from_numpy(noise[1]).cuda())
    assert sum(issubclass(w.category, Sr3RangeWarning) for w in rec) >= 2
    assert np.abs(out.cpu().numpy() - want).max() <= BAR
    want1 = oracle.p_sample(sd, cfg, oracle.noise_schedule(sched), noise[0], 7, cond, noise[1])
    assert np.abs(x1.cpu().numpy() - want1).max() <= BAR
    netG.denoise_fn.strict_range = True
    with pytest.raises(Sr3Error, match="fp16 range"):
        netG.super_resolution_batch(torch.from_numpy(cond).cuda(), noise=torch.from_numpy(noise))


def test_facade_is_ordered_against_torch_stream_by_events():
    """Inputs produced by torch kernels immediately before the call, outputs consumed by torch immediately after —
    on torch's default stream (the library runs on its own stream there) and on a side stream."""
    import torch
    P = pkg()
    cfg = synth.tiny_unet_config()
    sd = synth.synth_state_dict(cfg, 31)
    unet = P.UNet(in_channel=6, out_channel=3, inner_channel=cfg.inner_channel, norm_groups=cfg.norm_groups,
                  channel_mults=cfg.channel_mults, attn_res=cfg.attn_res, res_blocks=cfg.res_blocks,
                  image_size=cfg.image_size).cuda()
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rs = np.random.RandomState(4)
    x = rs.standard_normal((4, 6, 32, 32)).astype(np.float32)
    nl = rs.uniform(0.1, 0.9, (4, 1)).astype(np.float32)
    want = oracle.unet_forward(sd, cfg, x, nl)
    base = torch.from_numpy(x).cuda()
    big = torch.randn(4096, 4096, device="cuda")
    for stream in (None, torch.cuda.Stream()):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream()):
            for _ in range(3):
                _ = big @ big                              # keeps torch's stream busy in front of the input
                xin = (base * 2.0 - base) + 0.0            # the input is the product of queued torch kernels
                got = unet(xin, torch.from_numpy(nl).cuda())
                s = (got * 1.0).sum()                      # torch consumes the result without a host sync
            err = np.abs(got.cpu().numpy() - want).max()
            assert err <= 1e-4, err
            assert np.isfinite(float(s))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rccl_worker(port, q):
    """Child process: the nccl (= RCCL) process group is initialised before anything else touches the GPU."""
    import importlib
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", SR3_FORCE_COLLECTIVE="1")
    try:
        import torch
        import torch.distributed as dist
        name = "3d-super-resolution-face-reconstruction_amd"
        d = importlib.import_module(name + ".dist")
        sy = importlib.import_module(name + ".synth")
        sc = importlib.import_module(name + ".schedule")
        rank, world, local = d.init_from_env("nccl")
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
        cfg = sy.tiny_unet_config()
        eng = importlib.import_module(name + ".engine").Engine(cfg, 0)
        eng.load_state_dict(sy.synth_state_dict(cfg, 8))
        eng.set_precision("f16x3")
        eng.set_schedule(sc.schedule_buffers({"schedule": "linear", "n_timestep": 6, "linear_start": 1e-4, "linear_end": 2e-2}))
        n = 5
        x_full = torch.from_numpy(sy.synth_cond(n, 16, 8, 99))

        def sample_fn(x, off):
            out = torch.empty((x.shape[0], 3, 16, 16), dtype=torch.float32, device="cuda")
            xc = x.cuda().contiguous()
            torch.cuda.synchronize()
            eng.sample(xc.data_ptr(), x.shape[0], 16, 16, out.data_ptr(), None, 4321, off)
            eng.synchronize()
            return out

        local_out = d.sharded_super_resolution(sample_fn, x_full, gather=False)
        gathered = d.sharded_super_resolution(sample_fn, x_full)          # the collective branch (forced, one rank)
        same = bool(torch.equal(gathered, local_out)) and gathered.data_ptr() != local_out.data_ptr()
        # ragged branch of the same function (padded slices): with one rank it is the identity as well
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                          # bench.py's max-over-ranks timing reduce
        dist.barrier()
        q.put(("ok", same, float(t.item()), tuple(gathered.shape)))
        dist.destroy_process_group()
        eng.close()
    except Exception as ex:      # noqa: BLE001 — reported to the parent
        import traceback
        q.put(("error", traceback.format_exc(), repr(ex), None))


def test_rccl_collective_branch_runs_with_one_rank():
    """north_star's multi-GPU design has ONE collective; this executes it through RCCL (world size 1) with the
    product's own functions: dist.init_from_env("nccl"), dist.sharded_super_resolution -> all_gather_into_tensor.
    No scaling number comes out of it (no curve exists); it shows librccl initialises on this pool
    and the collective path is not dead code."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert p.exitcode == 0
    _, same, tmax, shape = res
    assert same and tmax == 1.5 and shape == (5, 3, 16, 16)
