"""Multi-process path on CPU (gloo, world_size 2 and 3): batch sharding + the single end-of-loop
all-gather + rank-invariant RNG offsets. The per-rank sampler is a stand-in function here (the HIP
sampler needs a GPU); the collective code is the product's (dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_sampler(x, off):
    # deterministic function of the image and its GLOBAL index, like the Philox image_offset
    idx = torch.arange(off, off + x.shape[0], dtype=torch.float32).view(-1, 1, 1, 1)
    return x * 0.5 + idx


def _worker(rank, world, port, n, q):
    import importlib, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    d = importlib.import_module("3d-super-resolution-face-reconstruction_amd.dist")
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d.init_from_env("gloo")
    x = torch.arange(n * 3 * 2 * 2, dtype=torch.float32).reshape(n, 3, 2, 2)
    out = d.sharded_super_resolution(_fake_sampler, x)
    a, b = d.shard_bounds(n, world, rank)
    q.put((rank, out.numpy(), (a, b)))
    dist.barrier()
    dist.destroy_process_group()


# (8, 512) and (8, 256): the global batches of BASELINE configs 4 and 5 (sr_sr3_VGGF2_8_128 / 32_128 on 8 GPUs: 64 and 32
# images per rank) — the 8-rank control flow of north_star's design, rehearsed over gloo on CPU; (8, 250): a ragged split
@pytest.mark.parametrize("world,n", [(2, 8), (2, 5), (3, 7), (8, 512), (8, 256), (8, 250)])
def test_sharded_gather(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.arange(n * 3 * 2 * 2, dtype=torch.float32).reshape(n, 3, 2, 2)
    want = _fake_sampler(x, 0).numpy()
    covered = []
    for rank, out, (a, b) in res:
        np.testing.assert_array_equal(out, want)      # every rank holds the full gathered batch
        covered += list(range(a, b))
    assert sorted(covered) == list(range(n))


def test_shard_bounds_properties():
    d = pkg("dist")
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            b = [d.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    d = pkg("dist")
    x = torch.randn(4, 3, 2, 2)
    assert torch.equal(d.sharded_super_resolution(lambda t, off: t + off, x), x)
    assert d.all_gather_images(x, 4) is x


class _FakeNet:
    """Stand-in for the torch facade on CPU: deterministic 'sampler' whose frame k of global image i is x_i + i + 100 k."""
    N_FRAMES = 10

    def parameters(self):
        return iter([torch.zeros(1)])

    def super_resolution_batch(self, x, seed=0, image_offset=0):
        return self.sample_batch(x, False, None, seed, image_offset)

    def sample_batch(self, x, continous=False, noise=None, seed=0, image_offset=0):
        idx = torch.arange(image_offset, image_offset + x.shape[0], dtype=torch.float32).view(-1, 1, 1, 1)
        frames = torch.stack([x + idx + 100.0 * k for k in range(self.N_FRAMES)], dim=0)
        return (frames[-1], frames) if continous else frames[-1]


def _worker_loop(rank, world, port, n, q):
    import importlib, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    d = importlib.import_module("3d-super-resolution-face-reconstruction_amd.dist")
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d.init_from_env("gloo")
    x = torch.arange(n * 3 * 2 * 2, dtype=torch.float32).reshape(n, 3, 2, 2)
    ret = d.sharded_p_sample_loop(_FakeNet(), x, continous=True, seed=5)
    last = d.sharded_p_sample_loop(_FakeNet(), x, continous=False, seed=5)
    q.put((rank, ret.numpy(), last.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 6), (3, 7)])
def test_sharded_p_sample_loop_returns_reference_ret_img(world, n):
    """continous=True across ranks: every rank gets the reference's `ret_img` (diffusion.py:203-215) — the conditioning
    batch, then the WHOLE global batch after every recorded step — and ret_img[-1]-style last image otherwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_loop, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.arange(n * 3 * 2 * 2, dtype=torch.float32).reshape(n, 3, 2, 2)
    _, frames = _FakeNet().sample_batch(x, True)
    want = torch.cat([x, frames.reshape(-1, 3, 2, 2)], dim=0).numpy()
    assert want.shape[0] == (1 + _FakeNet.N_FRAMES) * n
    for rank, ret, last in res:
        np.testing.assert_array_equal(ret, want)
        np.testing.assert_array_equal(last, want[-1])
