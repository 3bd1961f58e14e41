"""Checkpoint ingest: both on-disk layouts of the reference
(lib/trainer_temp.py:170-209,243-262; model/sr/model.py:139-195), written here with synthetic
weights (no pretrained file exists offline) and read back with `weights_only=True`."""
import numpy as np
import pytest

from conftest import cfg_from_meta, load_golden, pkg

synth = pkg("synth")
BAR = 1e-3


def _opt(cfg, sched):
    return {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": cfg.inner_channel,
                 "channel_multiplier": list(cfg.channel_mults), "attn_res": list(cfg.attn_res),
                 "res_blocks": cfg.res_blocks, "dropout": 0.0},
        "beta_schedule": {"train": sched, "val": sched},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": True}}}}


SCHED = {"schedule": "linear", "n_timestep": 20, "linear_start": 1e-4, "linear_end": 2e-2}


def _full_sd(cfg, seed, sched=SCHED):
    import torch
    schedule = pkg("schedule")
    sd = {"denoise_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, seed).items()}
    bufs = schedule.schedule_buffers(sched)
    for k in schedule.BUFFER_NAMES:
        sd[k] = torch.tensor(bufs[k], dtype=torch.float32)
    return sd


def test_gen_and_combined_layouts_round_trip(tmp_path):
    import torch
    ck = pkg("checkpoint")
    cfg = synth.tiny_unet_config()
    sd = _full_sd(cfg, 5)
    # upstream layout, addressed by prefix as the yml's pretrained_model_path does
    torch.save(sd, tmp_path / "I10_E1_gen.pth")
    # combined layout with DDP-style prefixes and the other entries the trainer writes
    torch.save({"sr_model_state": {"module." + k: v for k, v in sd.items()},
                "sr_optimizer_state": {"state": {}, "param_groups": []},
                "mica_model_state": {"w": torch.zeros(2)}, "mica_optimizer_state": {},
                "scheduler_state": {}, "epoch": 3, "global_step": 77, "batch_size_mica": 8},
               tmp_path / "I10_E1_checkpoint.pth")
    for target, layout, ep in [(str(tmp_path / "I10_E1"), "gen", None),
                               (str(tmp_path / "I10_E1_gen.pth"), "gen", None),
                               (str(tmp_path / "I10_E1_checkpoint.pth"), "combined", 3)]:
        net = pkg().define_G(_opt(cfg, SCHED))
        net.set_new_noise_schedule(SCHED, None)
        rep = ck.load_sr_checkpoint(net, target)
        assert rep.layout == layout and rep.epoch == ep
        assert rep.missing_keys == [] and rep.unexpected_keys == []
        assert rep.loaded == len(sd)
        got = net.state_dict()
        assert set(got) == set(sd)
        for k in sd:
            assert torch.equal(got[k].cpu(), sd[k]), k
    assert rep.global_step == 77


def test_strict_false_semantics_and_errors(tmp_path):
    import torch
    ck = pkg("checkpoint")
    cfg = synth.tiny_unet_config()
    sd = _full_sd(cfg, 6)
    # before set_new_noise_schedule the 12 buffers do not exist yet: they are "unexpected" and
    # ignored, exactly like nn.Module.load_state_dict(strict=False) on the reference class
    torch.save(sd, tmp_path / "a_gen.pth")
    net = pkg().define_G(_opt(cfg, SCHED))
    rep = ck.load_sr_checkpoint(net, str(tmp_path / "a"))
    assert sorted(rep.unexpected_keys) == sorted(pkg("schedule").BUFFER_NAMES) and not rep.missing_keys
    with pytest.raises(RuntimeError):
        ck.load_sr_checkpoint(net, str(tmp_path / "a"), strict=True)
    # a missing tensor and an extra one
    part = dict(sd)
    part.pop("denoise_fn.final_conv.block.3.bias")
    part["denoise_fn.extra.weight"] = torch.zeros(3)
    torch.save(part, tmp_path / "b_gen.pth")
    net.set_new_noise_schedule(SCHED, None)
    rep = ck.load_sr_checkpoint(net, str(tmp_path / "b"))
    assert rep.missing_keys == ["denoise_fn.final_conv.block.3.bias"]
    assert rep.unexpected_keys == ["denoise_fn.extra.weight"]
    # shape mismatch raises even with strict=False (torch semantics the reference relies on)
    bad = dict(sd)
    bad["denoise_fn.downs.0.weight"] = torch.zeros(32, 6, 1, 1)
    torch.save(bad, tmp_path / "c_gen.pth")
    with pytest.raises(RuntimeError):
        ck.load_sr_checkpoint(net, str(tmp_path / "c"))
    with pytest.raises(FileNotFoundError):
        ck.load_sr_checkpoint(net, str(tmp_path / "nope"))
    # a bare UNet takes the denoise_fn.* part
    unet = pkg().UNet(in_channel=6, out_channel=3, inner_channel=cfg.inner_channel, norm_groups=32,
                      channel_mults=cfg.channel_mults, attn_res=cfg.attn_res, res_blocks=cfg.res_blocks,
                      dropout=0.0, image_size=cfg.image_size)
    rep = ck.load_sr_checkpoint(unet, str(tmp_path / "a"))
    assert not rep.missing_keys and not rep.unexpected_keys
    assert torch.equal(unet.state_dict()["downs.0.weight"], sd["denoise_fn.downs.0.weight"])
    # files that are not tensor dicts are refused
    torch.save([1, 2, 3], tmp_path / "d_gen.pth")
    with pytest.raises(ValueError):
        ck.load_sr_checkpoint(net, str(tmp_path / "d"))
    # save_gen writes the upstream layout
    p = ck.save_gen(net, str(tmp_path / "e"))
    again = torch.load(p, weights_only=True)
    assert set(again) == set(net.state_dict())


@pytest.mark.gpu
def test_sampling_from_a_checkpoint_file_matches_the_reference_golden(tmp_path):
    """Weights AND schedule buffers come from the file (the model is built with a different
    schedule of the same length first): the sampler must use what was loaded, as the reference's
    buffer reads do (diffusion.py:144-162)."""
    import torch
    ck = pkg("checkpoint")
    g = load_golden("sampler_tiny.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    torch.save(_full_sd(cfg, m["seed"], m["schedule"]), tmp_path / "g_gen.pth")
    T = m["schedule"]["n_timestep"]
    netG = pkg().define_G(_opt(cfg, m["schedule"])).cuda()
    netG.set_new_noise_schedule(m["schedule"], [0])
    B, r = m["B"], m["r"]
    noise = torch.from_numpy(synth.synth_noise(T, B, 3, r, r, m["seed"]))
    cond = torch.from_numpy(g["cond"]).cuda()
    # perturb the buffers, sample (must differ), then ingest the file (must match the golden)
    with torch.no_grad():
        netG.posterior_mean_coef1.mul_(0.5)
    off = netG.p_sample_loop(cond, continous=False, noise=noise)
    assert np.abs(off.cpu().numpy() - g["last"]).max() > 10 * BAR
    rep = ck.load_sr_checkpoint(netG, str(tmp_path / "g"))
    assert not rep.missing_keys and not rep.unexpected_keys
    last = netG.p_sample_loop(cond, continous=False, noise=noise)
    assert np.abs(last.cpu().numpy() - g["last"]).max() <= BAR
