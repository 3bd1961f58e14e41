"""Round 4 (VERDICT r3 "next" items 1c, 4, 6, 7c and ADVICE r3):

  * BASELINE configs[4]'s per-GPU shard AT ITS OWN SIZE (32 images, 32 -> 128, T = 100, f16f8) against the reference's
    own run (tests/golden/sampler_cfg5_32_128.npz) — at B = 32 the fp8 path is ON at the 32x32 level and OFF at 16x16;
  * batches above the 4 GiB tensor limit run as equal chunks inside the facade (lib/trainer_temp.py:441-446: 15 samples
    x N images; configs[3]: 512 images) and the result does not depend on the chunking;
  * the in-place split-K conv with a CO-TENANT kernel holding the CU slots its sibling blocks need: a correct result and
    no hang (the bounded wait gives up, the library replays on the non-waiting path);
  * SR3_NO_HALO=1 in f16f8 mode (this combination used to abort the process);
  * two real ranks x 32 images with the fp8 path ON in every rank: what "any world size produces the same images" means
    in f16f8, as a number.

The reference computes in plain fp32 (model/sr/sr3_modules/unet.py:235-265, diffusion.py:164-215); bar 1e-3 (north_star).
"""
import ctypes
import os
import socket
import subprocess
import sys
import time
import warnings

import numpy as np
import pytest

import sr3_oracle as oracle
from conftest import REPO, cfg_from_meta, load_golden, pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")
schedule = pkg("schedule")
BAR = 1e-3


def _engine(cfg, sd, prec, sched=None):
    e = pkg("engine").Engine(cfg, 0)
    e.load_state_dict(sd)
    e.set_precision(prec)
    if sched is not None:
        e.set_schedule(schedule.schedule_buffers(sched))
    return e


def _opt(cfg, sched):
    return {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": cfg.in_channel, "out_channel": cfg.out_channel, "inner_channel": cfg.inner_channel,
                 "channel_multiplier": list(cfg.channel_mults), "attn_res": list(cfg.attn_res),
                 "res_blocks": cfg.res_blocks, "dropout": 0.0},
        "beta_schedule": {"train": sched, "val": sched},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": True}}}}


# ------------------------------------------------------------------------------------------------------------------
# config 5's shard at its own batch size
# ------------------------------------------------------------------------------------------------------------------
def test_cfg5_shard_b32_f16f8_vs_reference():
    """sr_sr3_VGGF2_32_128 on 8 GPUs = 32 images per GPU (BASELINE configs[4]): the reference's own T = 100 run (B = 2),
    replicated to exactly B = 32 with the same injected noise per replica pair, every replica held to the reference."""
    g = load_golden("sampler_cfg5_32_128.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    B0, r, T, st = m["B"], m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    B = 32
    rep = B // B0
    e = _engine(cfg, synth.synth_state_dict(cfg, m["seed"]), "f16f8", m["schedule"])
    # which convs take the fp8 path at this batch: the 32x32 level does, the 16x16 level (128 pixels x 32 images: too
    # few 128x128 tiles) does not — a different layer set than at B = 64
    assert e.conv_f8_supported(B, 32, 32, 256, 256) and e.conv_f8_supported(B, 32, 32, 256, 768)
    assert not e.conv_f8_supported(B, 16, 16, 512, 512)
    assert e.conv_f8_supported(64, 16, 16, 512, 512)
    noise = np.tile(synth.synth_noise(T, B0, 3, r, r, m["seed"]), (1, rep, 1, 1, 1))
    cond = np.tile(g["cond"], (rep, 1, 1, 1))
    final, frames = e.sample_np(cond, noise=noise, frames=True)
    assert e.fallback_calls() == 0 and e.replay_calls() == 0
    e.close()
    want_f = np.tile(g["frames_sub"], (1, rep, 1, 1, 1))
    err = np.abs(frames[..., ::st, ::st] - want_f).reshape(10, -1).max(1)
    e_fin = np.abs(final - np.tile(g["final"], (rep, 1, 1, 1))).max()
    print(f"cfg5 shard 32->128 T={T} B={B} [f16f8]: per-frame max abs err {np.array2string(err, precision=2)}; final {e_fin:.2e}")
    assert err.max() <= BAR and e_fin <= BAR
    np.testing.assert_array_equal(final[:B0], final[-B0:])


# ------------------------------------------------------------------------------------------------------------------
# batches above the per-call limit
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def net128():
    import torch
    cfg = synth.yml_unet_config(224)
    sched = {"schedule": "linear", "n_timestep": 2, "linear_start": 1e-4, "linear_end": 2e-2}
    netG = pkg().define_G(_opt(cfg, sched)).cuda()
    netG.load_state_dict({"denoise_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, 77).items()},
                         strict=False)
    netG.set_new_noise_schedule(sched, [0])
    yield netG
    netG.denoise_fn._engine.close()


def test_batch_of_300_is_chunked_and_chunking_is_invisible(net128):
    """B = 400 at 128x128 makes 4.8 GiB activation tensors (192 channels x 130 x 130 x 400): one sr3_sample call refuses it
    (sr3_max_batch = 330 for the yml UNet), the facade runs two equal chunks of 200. Rows must equal the same images sampled
    ALONE (image_offset = row) and the result must not depend on the number of chunks (f16x3: its products are batch
    independent up to fp32 summation order)."""
    import torch
    netG = net128
    netG.denoise_fn.precision = "f16x3"
    eng = netG._engine()                  # (engine with this net's weights and schedule pushed)
    limit = eng.max_batch(128, 128)
    assert 250 <= limit < 400, limit
    B, seed = 400, 4711
    assert netG.chunk_plan(B, limit) == (2, 200)
    # the library itself refuses the whole batch, naming the limit
    with pytest.raises(pkg("_lib").Sr3Error, match="at most"):
        big = torch.zeros((B, 3, 128, 128), device="cuda")
        out = torch.empty_like(big)
        eng.sample(big.data_ptr(), B, 128, 128, out.data_ptr(), None, seed, 0, None)
    cond = torch.from_numpy(synth.synth_cond(B, 128, 16, 31)).cuda()
    two = netG.super_resolution_batch(cond, seed=seed)
    three = netG.super_resolution_batch(cond, seed=seed, max_chunk=100)          # 4 chunks of 100
    ragged = netG.super_resolution_batch(cond[:295], seed=seed, max_chunk=100)   # 3 chunks of 99, the last one padded (97 + 2)
    assert two.shape == (B, 3, 128, 128) and torch.isfinite(two).all() and float(two.std()) > 0.2
    d23 = float((two - three).abs().max())
    d2r = float((two[:295] - ragged).abs().max())
    rows = [0, 199, 200, 399]
    alone = torch.cat([netG.super_resolution_batch(cond[i:i + 1], seed=seed, image_offset=i) for i in rows])
    d_alone = float((two[rows] - alone).abs().max())
    print(f"B=400 chunked: 2 vs 4 chunks {d23:.2e}; vs ragged 295 {d2r:.2e}; rows sampled alone {d_alone:.2e}")
    assert d23 <= 2e-6 and d2r <= 2e-6
    assert d_alone <= 2e-5
    # continous=True through the chunks: ret_img layout of the reference (x_in first, then every frame of the whole batch)
    out, frames = netG.sample_batch(cond, True, None, seed, 0, 100)
    assert frames.shape[:2] == (eng.num_frames(), B)
    assert float((out - two).abs().max()) <= 2e-6
    assert torch.equal(frames[-1], out)        # the last recorded frame is the step-0 image


def test_validation_of_15_samples_times_n_images_is_one_call(net128):
    """lib/trainer_temp.py:441-446 runs `cfg.sample` = 15 chains per validation image one at a time; validate_batch stacks
    them (sample k of image i = row k * N + i). 15 x 23 = 345 rows exceeds one library call (330): chunked transparently; the
    row / seed mapping is what a single-image call with image_offset = row produces."""
    import torch
    validation = pkg("validation")
    netG = net128
    netG.denoise_fn.precision = "f16x3"
    N, S, seed = 23, 15, 99
    sr = torch.from_numpy(synth.synth_cond(N, 128, 16, 5)).cuda()
    hr = torch.from_numpy(synth.synth_cond(N, 128, 64, 6)).cuda()
    res = validation.validate_batch(netG, sr, hr, samples=S, seed=seed)
    assert res["psnr"].shape == (S, N) and res["ssim"].shape == (S, N) and np.isfinite(res["mean_ssim"])
    imgs = res["images"]
    assert imgs.shape[0] == S * N
    assert netG.chunk_plan(S * N, netG._engine().max_batch(128, 128))[0] == 2
    for k, i in ((0, 0), (7, 3), (14, 22)):
        row = k * N + i
        alone = netG.super_resolution_batch(sr[i:i + 1], seed=seed, image_offset=row)
        assert float((imgs[row:row + 1] - alone).abs().max()) <= 2e-5


# ------------------------------------------------------------------------------------------------------------------
# in-place split-K with the CU slots taken by somebody else
# ------------------------------------------------------------------------------------------------------------------
def _filler():
    path = os.path.join(REPO, "tests", "gpu_helpers", "libsr3_test_filler.so")
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(path)
    lib.filler_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong]
    lib.filler_launch.restype = ctypes.c_int
    lib.filler_started.restype = ctypes.c_uint
    lib.filler_wait.restype = ctypes.c_int
    return lib


def test_inplace_splitk_with_cotenant_holding_the_cus():
    """The 8x8-level conv of the B = 64 step (M = 4096, 512 -> 512) runs on 128x128 x-halo tiles with the K range split
    over 4 blocks per tile that WAIT for each other (reduce-scatter tail): 512 blocks for the chip's 512 slots. Here a
    second stream holds almost every slot for 300 ms (510 holders of 60 KB LDS: two per CU, so no 73 KB conv block fits
    beside them; ~2 conv slots stay free): a block that gets a slot cannot see its siblings arrive, its bounded wait
    (5 ms) gives up, and the library replays the conv on the path without inter-block waits. Asserted: the call returns
    (no hang), the result equals the undisturbed run and the oracle, and the context keeps working afterwards."""
    Sr3ReplayWarning = pkg("_lib").Sr3ReplayWarning
    e = pkg("engine").Engine(synth.tiny_unet_config(), 0)
    e.load_state_dict(synth.synth_state_dict(e.cfg, 11))
    e.set_precision("f16x3")
    rs = np.random.RandomState(5)
    B, H, W, C, Co = 64, 8, 8, 512, 512
    x = rs.standard_normal((B, H, W, C)).astype(np.float32)
    w = (rs.standard_normal((Co, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rs.standard_normal(Co).astype(np.float32)
    calm = e.op_conv2d(x, w, bias)                       # undisturbed: the in-place split-K path
    assert e.replay_calls() == 0
    want = oracle.conv2d(x, w, bias)               # NHWC in / out
    assert np.abs(calm - want).max() < 2e-5
    fl = _filler()
    assert fl.filler_launch(510, 60 * 1024, 30_000_000) == 0       # 300 ms (op_conv2d packs its weights on the host first)
    t0 = time.time()
    while fl.filler_started() < 500 and time.time() - t0 < 5.0:
        time.sleep(0.0005)
    started = fl.filler_started()
    t1 = time.time()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        busy = e.op_conv2d(x, w, bias)
    dt = time.time() - t1
    assert fl.filler_wait() == 0
    replays = e.replay_calls()
    print(f"co-tenant: {started} holders resident; conv returned after {dt * 1e3:.1f} ms; replays {replays}; "
          f"warnings {[type(v.message).__name__ for v in wlist]}")
    assert dt < 5.0                                      # bounded: 300 ms of holders (~2 x 60 waits of 5 ms under them) + the replay
    assert np.abs(busy - want).max() < 2e-5
    assert np.abs(busy - calm).max() < 2e-5              # (split vs unsplit K order over K = 4608: fp32 summation order only)
    if replays:
        assert any(isinstance(v.message, Sr3ReplayWarning) for v in wlist)
        # the context stays on the non-waiting path: another disturbed conv needs no replay any more
        assert fl.filler_launch(510, 60 * 1024, 1_000_000) == 0
        again = e.op_conv2d(x, w, bias)
        assert fl.filler_wait() == 0
        assert e.replay_calls() == replays
        np.testing.assert_array_equal(again, busy)
    e.close()


def test_sampler_replays_the_segment_when_a_wait_gives_up():
    """sr3_sample's side of the same event, deterministically: a test kernel on another stream raises the 'wait gave up'
    bit of the context's flag word (sr3_test_flag_address) ~25 ms into a running call — what a timed-out split-K block does.
    The call must restore its last checkpoint, replay that segment on the non-waiting path IN THE SAME ARITHMETIC and
    return the images of an undisturbed call with SR3_OK_REPLAYED (Sr3ReplayWarning), not fail."""
    L = pkg("_lib")
    cfg = synth.yml_unet_config(224)
    sched = {"schedule": "linear", "n_timestep": 6, "linear_start": 1e-4, "linear_end": 2e-2}
    e = _engine(cfg, synth.synth_state_dict(cfg, 3), "f16f8", sched)
    cond = synth.synth_cond(64, 128, 16, 8)
    calm, calm_fr = e.sample_np(cond, seed=21, frames=True)
    assert e.replay_calls() == 0 and e.fallback_calls() == 0
    fl = _filler()
    fl.filler_poke.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
    fl.filler_poke.restype = ctypes.c_int
    addr = L.load().sr3_test_flag_address(e.ctx)
    assert addr
    # device buffers first (uploads synchronise), then the poke, then the call: 6 steps of ~17 ms
    dc = e.to_device(cond)
    out, fr = e.buffer(cond.size), e.buffer(e.num_frames() * cond.size)
    # The poke must land INSIDE the call (6 steps of ~17 ms); whether a kernel of another stream runs beside the call is the
    # machine's decision (hardware-queue assignment), so a poke that lands outside is tried again with another delay before
    # the test gives up as "not exercised on this box" — never a silent pass: a replay that happens is checked in full.
    for delay in (2_500_000, 4_500_000, 1_200_000):                # 25 ms, 45 ms, 12 ms
        assert fl.filler_poke(addr, 2, delay) == 0
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            e.sample(dc.ptr, 64, 128, 128, out.ptr, None, 21, 0, fr.ptr)
        assert fl.filler_wait() == 0
        e.synchronize()
        if e.replay_calls():
            break
        # (the bit landed before sample_begin's reset or after the last boundary: clear whatever is left of it)
        try:
            e.range_check()
        except L.Sr3Error:
            pass
    busy, busy_fr = out.download(cond.shape), fr.download((e.num_frames(),) + cond.shape)
    kinds = [type(v.message).__name__ for v in wlist]
    print(f"sampler with the wait flag raised mid-call: replays {e.replay_calls()}, fallbacks {e.fallback_calls()}, warnings {kinds}")
    if e.replay_calls() == 0:
        pytest.xfail("the test kernel never ran beside the sampler call on this box: replay path not exercised")
    assert e.replay_calls() == 1 and e.fallback_calls() == 0
    assert any(isinstance(v.message, L.Sr3ReplayWarning) for v in wlist)
    assert np.isfinite(busy).all()
    # same arithmetic; only the 8x8-level convs changed kernel (K summation order): fp32 round-off
    assert np.abs(busy - calm).max() <= 2e-5 and np.abs(busy_fr - calm_fr).max() <= 2e-5
    # the context stays on the non-waiting path and keeps working
    again = e.sample_np(cond, seed=21)
    assert e.replay_calls() == 1
    assert np.abs(again - busy).max() <= 2e-5       # (busy ran its first steps on the split-K path: K summation order)
    e.close()


# ------------------------------------------------------------------------------------------------------------------
# SR3_NO_HALO=1 with the f16f8 default (child process: the switch is read once per process)
# ------------------------------------------------------------------------------------------------------------------
_NO_HALO_CHILD = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, {root!r})
name = "3d-super-resolution-face-reconstruction_amd"
synth = importlib.import_module(name + ".synth")
Engine = importlib.import_module(name + ".engine").Engine
cfg = synth.yml_unet_config(224)
e = Engine(cfg, 0)
e.load_state_dict(synth.synth_state_dict(cfg, 9))
assert not e.conv_f8_supported(32, 32, 32, 256, 256)      # no x-halo kernel -> no fp8 path: ONE source of truth
B = 32
x = synth.synth_noise(1, B, 6, 128, 128, 4)[0]
nl = np.full((B,), 0.7, np.float32)
e.set_precision("f16f8")
got = e.unet_forward_np(x, nl)
e.set_precision("f32")
want = e.unet_forward_np(x, nl)
print("ERR", float(np.abs(got - want).max()), e.fallback_calls())
'''


def test_no_halo_switch_with_f16f8_default(tmp_path):
    """with SR3_NO_HALO=1 `conv_f8_supported` kept saying yes while `launch_conv` could not run the
    F8C kernel, and the host process was abort()ed. Now the switch is part of conv_f8_supported and launch_conv reports
    through the API's error path. A 128x128 forward at B = 32 (where the fp8 path would be on) in a child process."""
    script = tmp_path / "no_halo.py"
    script.write_text(_NO_HALO_CHILD.format(root=REPO))
    r = subprocess.run([sys.executable, str(script)], env={**os.environ, "SR3_NO_HALO": "1"}, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("ERR")][-1].split()
    print("SR3_NO_HALO=1, f16f8 forward vs f32:", line[1])
    assert float(line[1]) < 5e-4 and int(line[2]) == 0


# ------------------------------------------------------------------------------------------------------------------
# two ranks x 32 images: the fp8 path is ON in every rank
# ------------------------------------------------------------------------------------------------------------------
SCHED2 = {"schedule": "linear", "n_timestep": 3, "linear_start": 1e-4, "linear_end": 2e-2}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_worker(rank, world, port, n, q):
    import importlib
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    name = "3d-super-resolution-face-reconstruction_amd"
    P = importlib.import_module(name)
    d = importlib.import_module(name + ".dist")
    sy = importlib.import_module(name + ".synth")
    d.init_from_env("gloo")
    torch.cuda.set_device(0)
    cfg = sy.yml_unet_config(224)
    netG = P.define_G(_opt(cfg, SCHED2)).cuda()
    netG.load_state_dict({"denoise_fn." + k: torch.from_numpy(v) for k, v in sy.synth_state_dict(cfg, 515).items()}, strict=False)
    netG.set_new_noise_schedule(SCHED2, [0])
    assert netG.denoise_fn.precision == "f16f8"
    x_full = torch.from_numpy(sy.synth_cond(n, 128, 32, 99))
    ret = d.sharded_p_sample_loop(_CpuGather(netG), x_full, continous=True, seed=20261005)
    eng = netG.denoise_fn.engine()
    q.put((rank, ret.numpy(), bool(eng.conv_f8_supported(n // world, 32, 32, 256, 256)), eng.fallback_calls()))
    dist.barrier()
    dist.destroy_process_group()


class _CpuGather:
    """gloo has no CUDA collectives: hand the facade's results to the collective as host tensors."""

    def __init__(self, net):
        self.net = net

    def parameters(self):
        return self.net.parameters()

    def super_resolution_batch(self, x, seed=0, image_offset=0):
        return self.net.super_resolution_batch(x, seed=seed, image_offset=image_offset).cpu()

    def sample_batch(self, x, continous, noise, seed, image_offset):
        out, fr = self.net.sample_batch(x, continous, noise, seed, image_offset)
        return out.cpu(), fr.cpu()


def test_two_ranks_of_32_images_fp8_path_on():
    """2 ranks x 32 images at 128x128 (BASELINE configs[4]'s per-GPU shard), real HIP sampler in both, gloo gather of
    `continous=True` frames (ret_img layout). In f16f8 WHICH convs take the fp8 path depends on the rank's batch (B = 32:
    the 32x32 level; B = 64: also 16x16), so 'any world size gives the same images' holds to the fp8 correction error,
    not bit for bit: pinned here to a number (bar 1e-4; f16x3 / f32 are world-size independent to ~1e-6)."""
    import torch.multiprocessing as mp
    n, world = 64, 2
    cfg = synth.yml_unet_config(224)
    sd = synth.synth_state_dict(cfg, 515)
    cond = synth.synth_cond(n, 128, 32, 99)
    e = _engine(cfg, sd, "f16f8", SCHED2)
    want, wf = e.sample_np(cond, seed=20261005, frames=True)
    e.set_precision("f32")
    exact = e.sample_np(cond, seed=20261005)
    e.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    nf = wf.shape[0]
    for rank, ret, f8_on, fb in res:
        assert f8_on and fb == 0
        assert ret.shape == ((1 + nf) * n, 3, 128, 128)
        np.testing.assert_array_equal(ret[:n], cond)                       # x_in first (diffusion.py:203-204)
        frames = ret[n:].reshape(nf, n, 3, 128, 128)
        d_single = np.abs(frames[-1] - want).max()
        d_exact = np.abs(frames[-1] - exact).max()
        d_frames = np.abs(frames - wf).max()
        print(f"rank {rank}: 2 x 32 (f16f8) vs one process of 64 (f16f8) {d_single:.2e} (all frames {d_frames:.2e}); vs exact f32 {d_exact:.2e}")
        assert d_single <= 1e-4 and d_frames <= 1e-4 and d_exact <= 1e-4
    np.testing.assert_array_equal(res[0][1], res[1][1])                    # both ranks hold the same gathered ret_img


# ------------------------------------------------------------------------------------------------------------------
# the 64 -> 64 channel convs of the full-resolution level
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(16, 128, 128), (64, 128, 128), (64, 64, 64), (22, 96, 128), (33, 128, 128)])
def test_conv_64_to_64_full_resolution_vs_oracle(case):
    """3x3 / stride 1 / 64 -> 64 channels over >= 2048 tiles of 128 pixels (unet.py:80-91 `Block` conv at the 128x128
    level: 2.1 ms of the B = 64 step, the shapes furthest below the roofline). With bias and the per-image
    FeatureWiseAffine bias (unet.py:34-50), GroupNorm + Swish in front as the engine runs it; f16x3 bar as for every
    split-f16 conv: 2e-5 on O(1) outputs. The product library runs them on the x-halo kernel; the experiments build with
    SR3_WS=1 on the weights-stationary persistent kernel of profiles/README.md finding 66 (kernels_conv_ws.hip) — the same
    cases are its parity test (SR3_LIB=.../libsr3hip_exp.so SR3_WS=1 pytest -k conv_64_to_64)."""
    B, H, W = case
    e = pkg("engine").Engine(synth.tiny_unet_config(), 0)
    e.load_state_dict(synth.synth_state_dict(e.cfg, 11))
    e.set_precision("f16x3")
    rs = np.random.RandomState(B + H)
    x = rs.standard_normal((B, H, W, 64)).astype(np.float32)
    w = (rs.standard_normal((64, 64, 3, 3)) / np.sqrt(9 * 64)).astype(np.float32)
    bias = rs.standard_normal(64).astype(np.float32)
    cb = rs.standard_normal((B, 64)).astype(np.float32)
    sc = (1.0 + 0.1 * rs.standard_normal((B, 64))).astype(np.float32)
    sh = (0.1 * rs.standard_normal((B, 64))).astype(np.float32)
    got = e.op_conv2d(x, w, bias, gn_scale=sc, gn_shift=sh, swish=True, chan_bias=cb)
    act = oracle.swish(x * sc[:, None, None, :] + sh[:, None, None, :])
    want = oracle.conv2d(act, w, bias) + cb[:, None, None, :]
    err = np.abs(got - want).max()
    print(f"conv 64->64 B={B} {H}x{W}: max abs err {err:.2e}")
    assert err < 2e-5
    e.close()
