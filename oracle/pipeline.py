"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU fp32 restatement of the tensor part of
Pose2VideoPipeline.__call__ (src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:338-578)
and of the sliding-window scheduler (src/pipelines/context.py:7-42)."""
import math

import numpy as np
import torch

from .models import ReferenceAttentionControl


def ordered_halving(val):  # context.py:7-12
    bin_str = f"{val:064b}"
    return int(bin_str[::-1], 2) / (1 << 64)


def uniform(step, num_steps, num_frames, context_size, context_stride=3, context_overlap=4, closed_loop=True):
    """context.py:15-42 (generator -> list)."""
    if num_frames <= context_size:
        return [list(range(num_frames))]
    out = []
    context_stride = min(context_stride, int(np.ceil(np.log2(num_frames / context_size))) + 1)
    for context_step in 1 << np.arange(context_stride):
        pad = int(round(num_frames * ordered_halving(step)))
        for j in range(int(ordered_halving(step) * context_step) + pad,
                       num_frames + pad + (0 if closed_loop else -context_overlap),
                       (context_size * context_step - context_overlap)):
            out.append([e % num_frames for e in range(j, j + context_size * context_step, context_step)])
    return out


@torch.no_grad()
def denoise_clip(reference_unet, denoising_unet, scheduler, ehs, ref_latents, bk_latents, pose_fea, latents,
                 num_inference_steps, guidance_scale, context_frames=24, context_stride=1, context_overlap=4,
                 return_trajectory=False):
    """pipeline :373-374, :393-406, :462-564 on tensors.

    ehs: [1,1,768] CLIP embedding (cond); ref_latents [1,4,h,w]; bk_latents [1,4,F,h,w];
    pose_fea [1,320,F,h,w]; latents [1,4,F,h,w].  Returns final latents (and per-step latents)."""
    cfg = guidance_scale > 1.0
    scheduler.set_timesteps(num_inference_steps)
    if cfg:
        ehs = torch.cat([torch.zeros_like(ehs), ehs], dim=0)
    writer = ReferenceAttentionControl(reference_unet, "write", do_classifier_free_guidance=cfg)
    reader = ReferenceAttentionControl(denoising_unet, "read", do_classifier_free_guidance=cfg)
    rep = 2 if cfg else 1
    traj = []
    for i, t in enumerate(scheduler.timesteps):
        noise_pred = torch.zeros((latents.shape[0] * rep, *latents.shape[1:]), dtype=latents.dtype)
        counter = torch.zeros((1, 1, latents.shape[2], 1, 1), dtype=latents.dtype)
        if i == 0:
            reference_unet(ref_latents.repeat(rep, 1, 1, 1), torch.zeros_like(t), ehs)
            reader.update(writer)
        for c in uniform(0, num_inference_steps, latents.shape[2], context_frames, context_stride, context_overlap):
            x = latents[:, :, c].repeat(rep, 1, 1, 1, 1)
            x = torch.cat([x, bk_latents[:, :, c].repeat(rep, 1, 1, 1, 1)], dim=1)
            pose = pose_fea[:, :, c].repeat(rep, 1, 1, 1, 1)
            pred = denoising_unet(x, t, ehs[: x.shape[0]], pose_cond_fea=pose)
            noise_pred[:, :, c] = noise_pred[:, :, c] + pred
            counter[:, :, c] = counter[:, :, c] + 1
        if cfg:
            un, co = (noise_pred / counter).chunk(2)
            noise_pred = un + guidance_scale * (co - un)
        latents = scheduler.step(noise_pred, t, latents, eta=0.0).prev_sample
        traj.append(latents.clone())
    reader.clear()
    writer.clear()
    return (latents, traj) if return_trajectory else latents


@torch.no_grad()
def encode_image(vae, x):  # pipeline :427-431
    return vae.encode(x).latent_dist.mean * 0.18215


@torch.no_grad()
def decode_latents(vae, latents):  # pipeline :113-126
    f = latents.shape[2]
    z = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(-1, *latents.shape[1:2], *latents.shape[3:])
    video = torch.cat([vae.decode(z[i:i + 1]).sample for i in range(z.shape[0])])
    video = video.reshape(latents.shape[0], f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1).float()


@torch.no_grad()
def run_clip(vae, reference_unet, denoising_unet, pose_guider, scheduler, clip_embeds, ref_image, bk_images,
             pose_images, latents, num_inference_steps, guidance_scale, **ctx):
    """Whole tensor path of __call__: ref_image [1,3,H,W] in [-1,1], bk_images [F,3,H,W] in [-1,1],
    pose_images [F,3,H,W] in [0,1], clip_embeds [1,768], latents [1,4,F,h,w] -> videos [1,3,F,H,W] in [0,1]."""
    ref_lat = encode_image(vae, ref_image)
    bk = torch.stack([encode_image(vae, bk_images[i:i + 1])[0] for i in range(bk_images.shape[0])], dim=1)[None]
    pose_fea = pose_guider(pose_images.permute(1, 0, 2, 3)[None])
    lat = denoise_clip(reference_unet, denoising_unet, scheduler, clip_embeds[:, None], ref_lat, bk, pose_fea,
                       latents, num_inference_steps, guidance_scale, **ctx)
    return decode_latents(vae, lat), lat
