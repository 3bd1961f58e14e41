"""`GaussianDiffusion`: the reference's sampler API (model/sr/sr3_modules/diffusion.py:66-225)
over libsr3hip. The whole p_sample_loop — T UNet evaluations and the DDPM update — runs inside
the HIP library; this class only holds the schedule buffers (for state_dict parity) and moves
pointers. Training members (p_losses, forward, *_learn) are out of scope and raise.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import nn

from . import schedule as _schedule
from ._lib import Sr3Error


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, image_size, channels=3, loss_type="l1", conditional=True,
                 schedule_opt=None):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.conditional = conditional
        self.num_timesteps = 0
        self._sched_np = None
        self._sched_pushed = None   # engine id the schedule was pushed to

    # ---- reference surface that is configuration only -----------------------------------------
    def set_loss(self, device=None):
        # reference diffusion.py:85-91 builds an L1/L2 loss for training; nothing to do for sampling
        if self.loss_type not in ("l1", "l2"):
            raise NotImplementedError()

    def set_new_noise_schedule(self, schedule_opt, device=None):
        """diffusion.py:93-142. `device` may be 0, a list of ids (the reference's form), a
        torch.device or None (= where the denoiser lives)."""
        if isinstance(device, (list, tuple)):
            device = device[0]
        if device is None:
            device = next(self.denoise_fn.parameters()).device
        bufs = _schedule.schedule_buffers(schedule_opt)
        self.sqrt_alphas_cumprod_prev = bufs["sqrt_alphas_cumprod_prev"]
        self.num_timesteps = int(bufs["betas"].shape[0])
        for name in _schedule.BUFFER_NAMES:
            t = torch.tensor(bufs[name], dtype=torch.float32, device=device)
            if name in self._buffers:
                self._buffers[name] = t
            else:
                self.register_buffer(name, t)
        self._sched_np = bufs
        self._sched_pushed = None

    def _engine(self):
        eng = self.denoise_fn.engine()
        if self._sched_np is None:
            raise Sr3Error("set_new_noise_schedule() has not been called")
        # like the reference, sample with whatever the registered buffers hold now (a checkpoint's
        # load_state_dict may have replaced them, diffusion.py:144-162 reads self.<buffer>[t]); the
        # noise levels come from the config-derived float64 array (diffusion.py:108-109,166-167)
        sig = (id(eng),) + tuple((self._buffers[k].data_ptr(), self._buffers[k]._version)
                                 for k in _schedule.ENGINE_BUFFERS)
        if self._sched_pushed != sig:
            bufs = {"noise_level": self._sched_np["noise_level"]}
            for k in _schedule.ENGINE_BUFFERS:
                bufs[k] = self._buffers[k].detach().to("cpu", torch.float32).numpy()
            eng.set_schedule(bufs)
            self._sched_pushed = sig
        return eng

    # ---- sampling ------------------------------------------------------------------------------
    @staticmethod
    def _draw_seed() -> int:
        # one draw from torch's global generator: torch.manual_seed() makes sampling reproducible,
        # as with the reference's torch.randn calls (the streams themselves differ: Philox on device)
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    @staticmethod
    def chunk_plan(B: int, limit: int):
        """Equal chunks for a batch above the per-call limit: (n_chunks, chunk) with n_chunks * chunk >= B and the
        fewest chunks; the last one is PADDED to `chunk` images (rows discarded) so that every chunk runs the same
        workspace, the same captured graph and the same kernels (the tile rule depends on the batch)."""
        if B <= limit:
            return 1, B
        n = -(-B // limit)
        return n, -(-B // n)

    @torch.no_grad()
    def sample_batch(self, x_in, continous=False, noise: Optional[torch.Tensor] = None,
                     seed: Optional[int] = None, image_offset: int = 0, max_chunk: Optional[int] = None):
        """p_sample_loop for a whole batch (diffusion.py:189-215).

        x_in: conditioning [B,3,H,W] (conditional) or a shape tuple (unconditional, :193-201).
        noise: optional [T,B,C,H,W] tensor replacing the RNG (slab 0 = initial image, slab k = the
        randn_like of step t=T-k). Returns [B,C,H,W], or (final, frames [n,B,C,H,W]) if continous.

        Batches above the library's per-call limit (4 GiB per activation tensor: ~250 images at 128x128) — the
        reference's validation loop is 15 samples x N images (lib/trainer_temp.py:441-446), BASELINE configs[3] is 512
        images — run as equal chunks, one sr3_sample call each; Philox streams stay keyed by the GLOBAL image index
        (image_offset + row), so the result does not depend on the chunking. max_chunk lowers the limit (tests).
        """
        eng = self._engine()
        if self.conditional:
            x = x_in.to(torch.float32).contiguous()
            B, _, H, W = x.shape
            dev = x.device
        else:
            B, _, H, W = tuple(x_in)
            dev, x = next(self.denoise_fn.parameters()).device, None
        C = self.channels
        T = self.num_timesteps
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
        frames = None
        if continous:
            frames = torch.empty((eng.num_frames(), B, C, H, W), dtype=torch.float32, device=dev)
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            if tuple(noise.shape) != (T, B, C, H, W):
                raise RuntimeError(f"noise must be {(T, B, C, H, W)}, got {tuple(noise.shape)}")
        if seed is None:
            seed = self._draw_seed()
        limit = eng.max_batch(H, W)
        if max_chunk is not None:
            limit = max(1, min(limit, int(max_chunk)))
        n_chunks, chunk = self.chunk_plan(B, limit)
        if n_chunks == 1:
            self.denoise_fn.ready()
            eng.sample(x.data_ptr() if x is not None else None, B, H, W, out.data_ptr(),
                       noise.data_ptr() if noise is not None else None, seed, image_offset,
                       frames.data_ptr() if frames is not None else None)
            self.denoise_fn.finish()
            return (out, frames) if continous else out

        def rows(t, a, b, dim):
            """rows [a, b) of `t` along `dim`, padded to `chunk` rows by repeating the last one, contiguous"""
            part = t.narrow(dim, a, b - a)
            if b - a < chunk:
                last = part.narrow(dim, b - a - 1, 1)
                part = torch.cat([part] + [last] * (chunk - (b - a)), dim=dim)
            return part.contiguous()

        for k in range(n_chunks):
            a, b = k * chunk, min(B, (k + 1) * chunk)
            xc = rows(x, a, b, 0) if x is not None else None
            nc = rows(noise, a, b, 1) if noise is not None else None
            whole = b - a == chunk
            oc = out[a:b] if whole else torch.empty((chunk, C, H, W), dtype=torch.float32, device=dev)
            fc = torch.empty((frames.shape[0], chunk, C, H, W), dtype=torch.float32, device=dev) if continous else None
            self.denoise_fn.ready()
            eng.sample(xc.data_ptr() if xc is not None else None, chunk, H, W, oc.data_ptr(),
                       nc.data_ptr() if nc is not None else None, seed, image_offset + a,
                       fc.data_ptr() if fc is not None else None)
            self.denoise_fn.finish()
            if not whole:
                out[a:b] = oc[: b - a]
            if continous:
                frames[:, a:b] = fc[:, : b - a]
        return (out, frames) if continous else out

    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, noise=None, seed=None):
        """Reference return convention (diffusion.py:212-215): `ret_img` = cat([x_in, frames...])
        if continous else `ret_img[-1]` — the LAST image of the batch, shape [C,H,W]."""
        if not continous:
            return self.sample_batch(x_in, False, noise, seed)[-1]
        if seed is None:
            seed = self._draw_seed()
        out, frames = self.sample_batch(x_in, True, noise, seed)
        if self.conditional:
            first = x_in.to(torch.float32)
        elif noise is not None:
            # unconditional branch: ret_img starts with the initial noise image (diffusion.py:193-201,
            # `img = torch.randn(shape); ret_img = img`) = slab 0 of the injected noise
            first = noise[0].to(device=out.device, dtype=torch.float32)
        else:
            # device RNG: the initial image is draw 0 of every image's Philox stream
            # (init_state_kernel); regenerate it from the same (seed, image index) keys
            first = torch.empty_like(out)
            eng, n = self._engine(), out[0].numel()
            self.denoise_fn.ready()
            for i in range(out.shape[0]):
                eng.philox_normal_into(seed, i, 0, n, first[i].data_ptr())
            self.denoise_fn.finish()
        return torch.cat([first, frames.reshape(-1, *frames.shape[2:])], dim=0)

    @torch.no_grad()
    def sample(self, batch_size=1, continous=False):
        s = self.image_size
        return self.p_sample_loop((batch_size, self.channels, s, s), continous)

    @torch.no_grad()
    def super_resolution(self, x_in, continous=False):
        return self.p_sample_loop(x_in, continous)

    @torch.no_grad()
    def super_resolution_batch(self, x_in, noise=None, seed=None, image_offset=0, max_chunk=None):
        """Throughput entry point: every image of the batch, [B,3,H,W] (any B: see sample_batch)."""
        return self.sample_batch(x_in, False, noise, seed, image_offset, max_chunk)

    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=True, condition_x=None, noise=None):
        """One reverse step (diffusion.py:182-187) on the device."""
        if not clip_denoised:
            raise NotImplementedError("clip_denoised=False is never used by the reference")
        eng = self._engine()
        x = x.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        cond = condition_x.to(torch.float32).contiguous() if condition_x is not None else None
        if noise is None and t > 0:
            noise = torch.randn_like(x)
        nz = noise.to(torch.float32).contiguous() if (noise is not None and t > 0) else None
        out = torch.empty_like(x)

        def one_step():
            self.denoise_fn.ready()
            eng.sample_begin(cond.data_ptr() if cond is not None else None, B, H, W, x.data_ptr(), 0, 0)
            eng.sample_step(int(t), nz.data_ptr() if nz is not None else None)
            eng.sample_end(out.data_ptr())

        # The step API cannot replay inside the library (the caller owns the noise); this facade still holds x, cond and
        # noise, so it finishes the step like sr3_sample would: an in-place split-K wait that gave up -> the same step again
        # (the library has switched to its non-waiting conv path); an out-of-range step -> one arithmetic down, f16f8 ->
        # f16x3 -> exact f32 (the reference computes in fp32 and has no range limit, unet.py:235-265).
        import warnings
        from ._lib import Sr3RangeWarning, Sr3ReplayWarning
        prev = getattr(eng, "precision", "f32")
        ladder = {"f16f8": ["f16x3", "f32"], "f16x3": ["f32"], "f32": []}[prev]
        try:
            for attempt in range(8):
                try:
                    one_step()
                    break
                except Sr3Error as e:
                    msg = str(e)
                    if "repeat the call" in msg and attempt == 0:
                        warnings.warn(Sr3ReplayWarning(f"p_sample(t={int(t)}) repeated: {e}"), stacklevel=2)
                        continue
                    if "range" not in msg or not ladder or self.denoise_fn.strict_range:
                        raise
                    nxt = ladder.pop(0)
                    eng.set_precision(nxt)
                    warnings.warn(Sr3RangeWarning(f"p_sample(t={int(t)}) recomputed in {nxt}: {e}"), stacklevel=2)
        finally:
            if getattr(eng, "precision", prev) != prev:
                eng.set_precision(prev)
        self.denoise_fn.finish()
        return out

    # ---- small closed-form members kept for API completeness (diffusion.py:144-180) -------------
    def predict_start_from_noise(self, x_t, t, noise):
        return self.sqrt_recip_alphas_cumprod[t] * x_t - self.sqrt_recipm1_alphas_cumprod[t] * noise

    def q_posterior(self, x_start, x_t, t):
        mean = self.posterior_mean_coef1[t] * x_start + self.posterior_mean_coef2[t] * x_t
        return mean, self.posterior_log_variance_clipped[t]

    @torch.no_grad()
    def p_mean_variance(self, x, t, clip_denoised: bool, condition_x=None):
        B = x.shape[0]
        nl = torch.full((B, 1), float(np.float32(self.sqrt_alphas_cumprod_prev[t + 1])),
                        dtype=torch.float32, device=x.device)
        inp = torch.cat([condition_x, x], dim=1) if condition_x is not None else x
        x_recon = self.predict_start_from_noise(x, t, self.denoise_fn(inp, nl))
        if clip_denoised:
            x_recon.clamp_(-1.0, 1.0)
        return self.q_posterior(x_recon, x, t)

    # ---- training members: out of scope ---------------------------------------------------------
    def forward(self, x, *args, **kwargs):
        raise NotImplementedError("training loss (p_losses) is outside the SR3 sampling hot path")

    p_losses = q_sample = super_resolution_learn = p_sample_loop_learn = forward
