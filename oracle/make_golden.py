"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.safetensors by running the reference's OWN code
(/root/reference/src behind oracle/diffusers_standin.py, CPU fp32) on seeded synthetic weights and inputs.
Weights are NOT stored: tests rebuild them with oracle.synth.build (same seeds, same torch CPU generator).

    python -m oracle.make_golden [small] [forward512] [config1] [config2] [config2_video] [multiwindow]

  small       half-width UNets (4 heads, d = 40/80/160): denoising forward + banks, odd-size forward
  forward512  FULL-SIZE denoising UNet, config-2 shapes: one forward on 2 x 24 latent frames 64x64 (about 4 min, 15 GB)
  forward768  the same at BASELINE configs[4] shapes: 2 x 24 latent frames 96x96 (768x768; about 10 min, 30 GB)
  forward784  the scripts' DEFAULT size (run_animate.py:43-55: 784x784 -> latent 98x98, odd sizes 49 / 25 / 13 down the
              UNet, explicit-size upsampling on the way up): one denoising forward on 2 x 12 latent frames (about 6 min, 17 GB)
  config784   FULL-SIZE models at the default 784x784: 8 frames, 4 DDIM steps, CFG 3.5 (VAE encodes + pose guider + both UNets)
  vae784      one frame of the VAE decoder at 784x784 (latent 98x98) with the oracle VAE (diffusers AutoencoderKL restatement)
  config1     FULL-SIZE models, BASELINE config 1: 256x256, 8 frames, 4 DDIM steps, CFG 3.5 (latents after every step)
  config2     FULL-SIZE models, BASELINE config 2: 512x512, 24 frames, 20 DDIM steps, CFG 3.5 (latents after steps
              0, 9, 19; about 70 min and 19 GB on 8 cores)
  config2_video  decoded frames 0 and 23 of config2's final latents (oracle VAE decode of the stored reference latents)
  multiwindow FULL-SIZE models through the reference's OWN Pose2VideoPipeline.__call__ (PIL in, .videos out):
              512x512, F = 48 -> three 24-frame context windows (the last wraps to frame 0), 4 DDIM steps, CFG 3.5;
              latents after steps 0 and 3 (callback), decoded frames 0 and 47 (about 40 min and 25 GB on 8 cores)
"""
import os
import sys
import time

import torch
from safetensors.torch import save_file as _save_file

from . import models as OM
from . import primitives as OP
from . import synth
from .diffusers_standin import install

def save_file(tensors, path):
    _save_file({k: v.detach().float().contiguous() for k, v in tensors.items()}, path)


OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

MM = dict(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
          use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
          motion_module_decoder_only=False, motion_module_type="Vanilla")


def mm_kwargs(heads):
    return dict(MM, motion_module_kwargs=dict(num_attention_heads=heads, num_transformer_block=1,
                                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                              temporal_attention_dim_div=1))


def ref_models(kw, heads, hw, seed3, seed2):
    """Reference-code UNets loaded with the seeded oracle weights (strict)."""
    import src.models.unet_2d_condition as u2
    import src.models.unet_3d_edit_bkfill as u3
    o3 = synth.build(OM.UNet3DConditionModel, seed3, motion_heads=heads, **kw)
    o2 = synth.build(OM.UNet2DConditionModel, seed2, **kw)
    r3 = u3.UNet3DConditionModel(sample_size=hw, in_channels=8, **kw, **mm_kwargs(heads)).eval()
    r2 = u2.UNet2DConditionModel(sample_size=hw, in_channels=4, **kw).eval()
    r3.load_state_dict(o3.state_dict(), strict=True)
    r2.load_state_dict(o2.state_dict(), strict=True)
    del o3, o2
    return r3, r2


def forward_case(r3, r2, hw, F, C0, seed, t=749):
    import src.models.mutual_self_attention as msa
    g = torch.Generator().manual_seed(seed)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    ref_lat = torch.randn(1, 4, hw, hw, generator=g)
    x = torch.randn(2, 8, F, hw, hw, generator=g)
    pose = torch.randn(2, C0, F, hw, hw, generator=g)
    with torch.no_grad():
        w = msa.ReferenceAttentionControl(r2, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
        rd = msa.ReferenceAttentionControl(r3, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
        r2(ref_lat.repeat(2, 1, 1, 1), torch.zeros(()), encoder_hidden_states=ehs, return_dict=False)
        rd.update(w)
        out = r3(x, torch.tensor(t), encoder_hidden_states=ehs, pose_cond_fea=pose, return_dict=False)[0]
        rd.clear()
        w.clear()
    return out


def small():
    kw = synth.small_unet_kwargs()
    r3, r2 = ref_models(kw, 4, 16, 31, 32)
    out = {"fwd_hw16_F8": forward_case(r3, r2, 16, 8, 160, 6), "fwd_hw13_F3": forward_case(r3, r2, 13, 3, 160, 6)}
    save_file(out, os.path.join(OUT, "small_unet_forward.safetensors"))
    print("small:", {k: tuple(v.shape) for k, v in out.items()})


def forward512():
    t0 = time.time()
    r3, r2 = ref_models(OM.SD15_UNET_CONFIG, 8, 64, 1234, 1235)
    print(f"full-size reference models built in {time.time()-t0:.0f} s")
    t0 = time.time()
    out = forward_case(r3, r2, 64, 24, 320, 9, t=499)
    print(f"full-size forward {time.time()-t0:.0f} s")
    save_file({"fwd_hw64_F24": out}, os.path.join(OUT, "full_unet_forward_512.safetensors"))


def forward768():
    """BASELINE configs[4] shapes: one denoising forward at 768x768 (latent 96x96), 2 x 24 frames, reference bank."""
    t0 = time.time()
    r3, r2 = ref_models(OM.SD15_UNET_CONFIG, 8, 96, 1234, 1235)
    print(f"full-size reference models built in {time.time()-t0:.0f} s", flush=True)
    t0 = time.time()
    out = forward_case(r3, r2, 96, 24, 320, 13, t=499)
    print(f"768x768 forward {time.time()-t0:.0f} s", flush=True)
    save_file({"fwd_hw96_F24": out}, os.path.join(OUT, "full_unet_forward_768.safetensors"))


def forward784():
    """run_animate.py / run_edit.py default size 784x784 (run_animate.py:43-55): latent 98 -> 49 -> 25 -> 13, so every
    down-sampler sees an odd size and every up-sampler takes the explicit `upsample_size` path (unet_3d_edit_bkfill.py)."""
    t0 = time.time()
    r3, r2 = ref_models(OM.SD15_UNET_CONFIG, 8, 98, 1234, 1235)
    print(f"full-size reference models built in {time.time()-t0:.0f} s", flush=True)
    t0 = time.time()
    out = forward_case(r3, r2, 98, F784, 320, 17, t=499)
    print(f"784x784 forward {time.time()-t0:.0f} s", flush=True)
    save_file({f"fwd_hw98_F{F784}": out}, os.path.join(OUT, "full_unet_forward_784.safetensors"))


F784 = 12


def vae784():
    """One decoded frame at 784x784: decode_latents (pipeline :113-126) of a seeded 98x98 latent with the oracle VAE."""
    vae = synth.build(OP.AutoencoderKL, 1237)
    g = torch.Generator().manual_seed(19)
    lat = torch.randn(1, 4, 98, 98, generator=g)
    with torch.no_grad():
        img = vae.decode(lat / 0.18215).sample
        enc = vae.encode(img.clamp(-1, 1)).latent_dist.mean * 0.18215
    save_file({"latent": lat, "video_frame": (img / 2 + 0.5).clamp(0, 1)[0], "reencoded_latent": enc},
              os.path.join(OUT, "vae_784_frame.safetensors"))
    print("vae784 done", tuple(img.shape), flush=True)


def config1():
    """BASELINE configs[0]: 256x256, 8 frames, 4 DDIM steps, CFG 3.5 through the reference's own models."""
    _clip(256, 8, 4, range(4), "config1_256_8f_4steps.safetensors")


def config784():
    """The scripts' default frame size (run_animate.py:43-55: 784x784) through the whole tensor path: VAE encode of the
    reference image and of 8 background frames, pose guider, reference UNet, 4 DDIM steps with CFG on 98x98 latents (odd sizes
    49 / 25 / 13 down the UNet); latents after every step (about 15 min on 8 cores)."""
    _clip(784, 8, 4, range(4), "config784_8f_4steps.safetensors")


def config2():
    """BASELINE configs[1] (the bench workload): 512x512, 24 frames (one context window), 20 DDIM steps, CFG 3.5."""
    _clip(512, 24, 20, (0, 9, 19), "config2_512_24f_20steps.safetensors", keep_pose=False)


def _clip(size, F, steps, keep, fname, keep_pose=True):
    from src.pipelines.pipeline_pose2vid_long_edit_bkfill_roiclip import Pose2VideoPipeline
    import src.models.pose_guider as pg
    r3, r2 = ref_models(OM.SD15_UNET_CONFIG, 8, size // 8, 1234, 1235)
    opg = synth.build(OM.PoseGuider, 1236)
    rpg = pg.PoseGuider(320, 3, (16, 32, 96, 256)).eval()
    rpg.load_state_dict(opg.state_dict())
    vae = synth.build(OP.AutoencoderKL, 1237)
    from .pipeline import run_clip  # tensor-level driver of the same models (proven equal to __call__ in tests)

    class _Wrap(torch.nn.Module):  # adapt reference-module call signatures to the tensor driver
        def __init__(s, m):
            super().__init__()
            s.m = m
    # The reference pipeline works on PIL images; to inject exact tensors we drive the reference MODELS with the
    # oracle's tensor-level loop (tests/test_oracle_vs_reference.py::test_pipeline_equals_reference proves that loop
    # equal to Pose2VideoPipeline.__call__).
    import src.models.mutual_self_attention as msa
    from .pipeline import uniform
    H = W = size
    gs = 3.5
    g = torch.Generator().manual_seed(11)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.ones(F, 3, H, W)
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    sched = OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS)
    sched.set_timesteps(steps)
    with torch.no_grad():
        ref_lat = vae.encode(ref_img).latent_dist.mean * 0.18215
        bk_lat = torch.stack([(vae.encode(bk[i:i + 1]).latent_dist.mean * 0.18215)[0] for i in range(F)], dim=1)[None]
        pose_fea = rpg(pose.permute(1, 0, 2, 3)[None])
        ehs = torch.cat([torch.zeros(1, 1, 768), clip[:, None]])
        w = msa.ReferenceAttentionControl(r2, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
        rd = msa.ReferenceAttentionControl(r3, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
        traj = {}
        for i, t in enumerate(sched.timesteps):
            if i == 0:
                r2(ref_lat.repeat(2, 1, 1, 1), torch.zeros_like(t), encoder_hidden_states=ehs, return_dict=False)
                rd.update(w)
            c = list(range(F))
            x = torch.cat([lat.repeat(2, 1, 1, 1, 1), bk_lat.repeat(2, 1, 1, 1, 1)], dim=1)
            pred = r3(x, t, encoder_hidden_states=ehs, pose_cond_fea=pose_fea.repeat(2, 1, 1, 1, 1), return_dict=False)[0]
            un, co = pred.chunk(2)
            lat = sched.step(un + gs * (co - un), t, lat, eta=0.0).prev_sample
            if i in keep:
                traj[f"latents_step{i}"] = lat.clone()
            print("step", i, int(t), time.strftime("%H:%M:%S"), flush=True)
    traj["ref_latents"] = ref_lat
    if keep_pose:
        traj["pose_fea_frame0"] = pose_fea[:, :, 0].contiguous()
    save_file(traj, os.path.join(OUT, fname))


def config2_video():
    """Decoded video frames for the config-2 fixture: the oracle VAE (diffusers AutoencoderKL restatement, seed 1237)
    applied to the reference's final latents as decode_latents does (pipeline :113-126)."""
    from safetensors.torch import load_file
    G = load_file(os.path.join(OUT, "config2_512_24f_20steps.safetensors"))
    vae = synth.build(OP.AutoencoderKL, 1237)
    lat = G["latents_step19"] / 0.18215
    out = {}
    with torch.no_grad():
        for f in (0, 23):
            img = vae.decode(lat[:, :, f]).sample
            out[f"video_frame{f}"] = (img / 2 + 0.5).clamp(0, 1)[0]
    save_file(out, os.path.join(OUT, "config2_512_24f_video_frames.safetensors"))
    print("config2_video:", {k: tuple(v.shape) for k, v in out.items()})


MW = dict(size=512, F=48, steps=4, guidance=3.5, seed=42)


def multiwindow_inputs(size, F):
    """Synthetic PIL inputs of the multi-window fixture (regenerated by the test, not stored)."""
    import numpy as np
    from PIL import Image
    mk = lambda s: Image.fromarray(np.random.RandomState(s).randint(0, 256, (size, size, 3), dtype=np.uint8))
    return mk(0), [mk(100 + i) for i in range(F)], [mk(200 + i) for i in range(F)]


def clip_embedding():
    return torch.randn(1, 768, generator=torch.Generator().manual_seed(3))


class FakeClip(torch.nn.Module):
    """image_encoder stub: a fixed seeded embedding (the CLIP tower is pinned separately against transformers)."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.emb = clip_embedding()

    @property
    def dtype(self):
        return torch.float32

    def forward(self, x):
        return type("O", (), {"image_embeds": self.emb.to(x.device)})()


def multiwindow():
    from src.pipelines.pipeline_pose2vid_long_edit_bkfill_roiclip import Pose2VideoPipeline
    import src.models.pose_guider as pg
    size, F, steps = MW["size"], MW["F"], MW["steps"]
    t0 = time.time()
    r3, r2 = ref_models(OM.SD15_UNET_CONFIG, 8, size // 8, 1234, 1235)
    opg = synth.build(OM.PoseGuider, 1236)
    rpg = pg.PoseGuider(320, 3, (16, 32, 96, 256)).eval()
    rpg.load_state_dict(opg.state_dict())
    vae = synth.build(OP.AutoencoderKL, 1237)
    sched = OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS)
    pipe = Pose2VideoPipeline(vae=vae, image_encoder=FakeClip(), reference_unet=r2, denoising_unet=r3, pose_guider=rpg,
                              scheduler=sched)
    print(f"models built in {time.time()-t0:.0f} s", flush=True)
    ref_img, poses, bks = multiwindow_inputs(size, F)
    traj = []

    def cb(i, t, lat):  # the reference passes the shadowed window-batch index as i (:505); count calls instead
        traj.append(lat.detach().clone())
        print("step", len(traj) - 1, int(t), time.strftime("%H:%M:%S"), flush=True)

    video = pipe(ref_img, poses, bks, size, size, F, steps, MW["guidance"], generator=torch.manual_seed(MW["seed"]),
                 callback=cb, callback_steps=1).videos
    assert len(traj) == steps and video.shape == (1, 3, F, size, size)
    out = {"latents_step0": traj[0], f"latents_step{steps-1}": traj[-1],
           "video_frame0": video[0, :, 0], f"video_frame{F-1}": video[0, :, F - 1]}
    save_file(out, os.path.join(OUT, "multiwindow_512_48f_4steps.safetensors"))
    print(f"multiwindow done in {time.time()-t0:.0f} s", {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    install()
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1:] or ["small"]
    for w in what:
        {"small": small, "forward512": forward512, "forward768": forward768, "forward784": forward784, "vae784": vae784, "config784": config784, "config1": config1, "config2": config2,
         "config2_video": config2_video, "multiwindow": multiwindow}[w]()
