#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native MIMO denoising path.

Metric (BASELINE.json): denoised frames/sec for a 512x512, 24-frame clip, 20 DDIM steps, CFG 3.5 —
one "step" of this script = ONE whole clip through Pose2VideoPipeline.run_tensors (VAE encode of the
reference + 24 background frames, pose guider, reference UNet, 20 x {denoising UNet (48 images), window
average, guidance, DDIM}, VAE decode of 24 frames) with the inputs already resident in HBM.
Synthetic inputs and seeded random-init weights of the real architecture (no checkpoints offline).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

N > 1 (weak scaling): every rank denoises its own independent 24-frame clip (clips are independent objects:
no data-path collective); value = N * 24 frames / max-over-ranks time.  `--shard-windows` instead runs ONE
long clip of 24*N frames whose (window, CFG-half) units are dealt over the ranks with one all_gather per step
(BASELINE config 4).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (denoising-UNet forward, the dominant
launch sequence: algorithmic FLOPs actually executed / HIP-event time vs the dense 16-bit MFMA peak) and
`cpu_baseline` (the CPU fp32 oracle — a port of the reference's PyTorch path — timed on this host's cores
on a bounded sample and extrapolated by FLOPs).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
NOISE_SCHEDULER_KWARGS = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                              steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                              timestep_spacing="trailing")


def randomize_(module, seed):
    """Synthetic weights: default inits, zero-initialised tensors re-drawn N(0, 0.02^2),
    norm affines 1 + 0.1 N / 0.1 N, so that no branch is numerically invisible."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            elif isinstance(m, (nn.Conv2d, nn.Linear)) and float(m.weight.abs().max()) == 0.0:
                m.weight.copy_(0.02 * torch.randn(m.weight.shape, generator=g))
    return module


def build_pipeline(dev, dtype, seed=1234):
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from mimo_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    from mimo_amd.vae import AutoencoderKL, PoseGuider
    from mimo_amd.clip import CLIPVisionModelWithProjection
    torch.manual_seed(seed)
    with torch.device(dev):  # parameters are initialised directly in HBM
        den = UNet3DConditionModel()
        ref = UNet2DConditionModel()
        vae = AutoencoderKL()
        pg = PoseGuider()
        clip = CLIPVisionModelWithProjection()  # ViT-L/14 vision tower + projection (pretrained_weights/image_encoder)
    for i, m in enumerate((den, ref, vae, pg, clip)):
        randomize_(m, seed + 1 + i)
        m.to(dtype=dtype)  # weights held in the compute dtype, like the reference's weight_dtype
        m.compute_dtype = dtype
        m.requires_grad_(False)
    return Pose2VideoPipeline(vae, clip, ref, den, pg, DDIMScheduler(**NOISE_SCHEDULER_KWARGS))


def synthetic_inputs(dev, frames, size, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    h = size // 8
    return dict(ref_image=(torch.rand(1, 3, size, size, generator=g) * 2 - 1).to(dev),
                bk_images=torch.ones(frames, 3, size, size, device=dev),  # run_animate: white background (tools/util.py:339-345)
                pose_images=torch.rand(frames, 3, size, size, generator=g).to(dev),
                clip_embeds=torch.randn(1, 768, generator=g).to(dev),
                clip_pixels=torch.randn(1, 3, 224, 224, generator=g).to(dev),  # CLIPImageProcessor output of the reference image
                latents=torch.randn(1, 4, frames, h, h, generator=g).to(dev))


def measure_forward(pipe, dev, dtype, size, iters=3):
    """HIP-event timing (current stream = the launch stream of every kernel) of ONE denoising-UNet forward on a
    CFG batch of 2 x 24 latent frames with banks installed, plus the algorithmic FLOPs it executes."""
    from mimo_amd import ops
    from mimo_amd.modules import Ctx, EarlyExit
    from mimo_amd.unet import ReferenceAttentionControl
    h = size // 8
    unet, refu = pipe.denoising_unet, pipe.reference_unet
    g = torch.Generator(device="cpu").manual_seed(7)
    writer = ReferenceAttentionControl(refu, mode="write", do_classifier_free_guidance=True)
    reader = ReferenceAttentionControl(unet, mode="read", do_classifier_free_guidance=True)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)]).to(dev)
    rctx = Ctx(dtype, 1, 1)
    rctx.stop_after = writer.last_block()
    try:
        refu.run_tokens(torch.randn(1, h, h, 8, generator=g).to(dev).to(dtype), 0, ehs[1:], 1, 1, None, rctx)
    except EarlyExit:
        pass
    reader.update(writer)
    x = torch.randn(48, h, h, 8, generator=g).to(dev).to(dtype)
    pose = torch.randn(48, h, h, 320, generator=g).to(dev)
    ops.COUNTER = {"flops": 0, "launches": 0}
    unet.run_tokens(x, 499, ehs, 2, 24, pose)
    flops, launches = ops.COUNTER["flops"], ops.COUNTER["launches"]
    ops.COUNTER = None
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        unet.run_tokens(x, 499, ehs, 2, 24, pose)
    en.record()
    torch.cuda.synchronize()
    # dominant kernel family: every gemm_kernel launch (implicit-GEMM convs + linears) of ONE forward bracketed by
    # HIP events on the launch stream
    ops.EVENTS = []
    unet.run_tokens(x, 499, ehs, 2, 24, pose)
    torch.cuda.synchronize()
    fam = {}
    for name, e0, e1, fl, nb in ops.EVENTS:
        d = fam.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0, "bytes": 0})
        d["launches"] += 1
        d["ms"] += e0.elapsed_time(e1)
        d["flops"] += fl
        d["bytes"] += nb
    ops.EVENTS = None
    reader.clear()
    writer.clear()
    return st.elapsed_time(en) / iters * 1e-3, flops, launches, fam


def lib_hash():
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "mimo_amd", "libmimo_hip.so"), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(family):
    """HBM-side bytes per launch of a kernel family from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    over tools/profile_forward.py (same models, same shapes; PMC passes cannot run inside the timed region).
    None when no summary has been committed for this build (tools/pmc_traffic.py writes it)."""
    for name in ("r6_pmc_forward_traffic.json", "r5_pmc_forward_traffic.json", "r4_pmc_forward_traffic.json", "r3_pmc_forward_traffic.json", "r2_pmc_forward_traffic.json", "r1_pmc_forward_traffic.json"):  # newest first
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            d = json.load(open(path))[family]
            full = json.load(open(path))
            return {"bytes_per_launch": d["traffic_bytes_per_launch"], "fetch": d["fetch_bytes_per_launch"],
                    "write": d["write_bytes_per_launch"], "source": "profiles/" + name,
                    "library_sha256_16": full.get("library_sha256_16"), "library_sha256_16_now": lib_hash(),
                    "note": "family = every kernel bench.py brackets as gemm_kernel; summaries stamped with a library older than "
                            "b95d744fdbf16a78 missed gemm8_kernel (DESIGN.md section 5)"}
        except Exception:
            continue
    return None


def cpu_baseline(clip_flops, frames, budget_s=25.0, config2=False):
    """The reference's CPU PyTorch path timed on this host: ONE denoising-UNet forward of the FULL-SIZE model at
    BASELINE configs[0] size (latent 32x32, 2 x 8 frames: ~3.3 TFLOP, seconds per forward), FLOPs counted by torch's
    FlopCounterMode, extrapolated to the whole clip's executed FLOPs.  Where /root/reference is mounted (this
    container) the model is the reference's OWN src/models code behind oracle/diffusers_standin.py (kind "reference");
    on the GPU box, where it is not, the oracle port proven equal to it (kind "port")."""
    from torch.utils.flop_counter import FlopCounterMode
    from oracle import models as OM
    cores = min(os.cpu_count(), 32)  # threads actually used: more only adds oversubscription on shared hosts
    torch.set_num_threads(cores)
    t0 = time.time()
    kind = "port"
    m = None
    if os.path.isdir("/root/reference/src"):
        try:
            from oracle.diffusers_standin import install
            from oracle.make_golden import mm_kwargs
            install()
            import src.models.unet_3d_edit_bkfill as u3
            with torch.device("meta"):
                m = u3.UNet3DConditionModel(sample_size=64, in_channels=8, **OM.SD15_UNET_CONFIG, **mm_kwargs(8))
            kind = "reference"
        except Exception:
            m = None
    if m is None:
        with torch.device("meta"):
            m = OM.UNet3DConditionModel()
    m = m.to_empty(device="cpu")
    with torch.no_grad():
        for p in m.parameters():
            p.uniform_(-0.03, 0.03)
        for mod in m.modules():
            if isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
                mod.weight.fill_(1.0)
                mod.bias.zero_()
            if hasattr(mod, "pe") and torch.is_tensor(getattr(mod, "pe")):  # PositionalEncoding buffer
                mod.pe.copy_(OM.PositionalEncoding(mod.pe.shape[-1], mod.pe.shape[1]).pe)
    m.eval()
    build_s = time.time() - t0
    hw, F = (64, 24) if config2 else (32, 8)
    x = torch.randn(2, 8, F, hw, hw)
    ehs = torch.randn(2, 1, 768)
    pose = torch.randn(2, 320, F, hw, hw)

    def fwd():
        if kind == "reference":
            return m(x, torch.tensor(499), encoder_hidden_states=ehs, pose_cond_fea=pose, return_dict=False)[0]
        return m(x, torch.tensor(499), ehs, pose_cond_fea=pose)

    with torch.no_grad():
        if config2:  # one forward is about a minute: the counted pass IS the timed pass (no warm-up)
            t1 = time.time()
            with FlopCounterMode(display=False) as fc:
                fwd()
            dt, n = time.time() - t1, 1
            sample_flops = fc.get_total_flops()
        else:
            with FlopCounterMode(display=False) as fc:
                fwd()  # warm-up + FLOP count
            sample_flops = fc.get_total_flops()
            n, t1 = 0, time.time()
            while n < 1 or (time.time() - t1 < budget_s and n < 5):
                fwd()
                n += 1
            dt = (time.time() - t1) / n
    cpu_flops_per_s = sample_flops / dt
    what = "the reference's src/models UNet3DConditionModel (oracle/diffusers_standin.py)" if kind == "reference" \
        else "oracle.models.UNet3DConditionModel (port of the reference's PyTorch path)"
    return dict(value=frames / (clip_flops / cpu_flops_per_s), unit="frames/s", cores=cores, kind=kind,
                sample=(f"{what}, full size, fp32, {cores} threads: denoising forward on 2x{F} frames at latent {hw}x{hw} "
                        f"(BASELINE configs[{1 if config2 else 0}] size), {n} timed: {dt:.2f} s/forward, {sample_flops/1e12:.3f} TFLOP => "
                        f"{cpu_flops_per_s/1e12:.3f} TFLOP/s; extrapolated to the clip's {clip_flops/1e12:.1f} executed TFLOP "
                        f"(model build {build_s:.0f} s untimed); a single un-warmed sample: read it as ~{frames / (clip_flops / cpu_flops_per_s):.3f} "
                        f"+- 8 % (0.0135-0.0157 over the runs of rounds 4-6)"
                        + ("" if config2 else "; QUICK sample: at the workload's own configs[1] shape a 32-thread MI355X host sustains "
                           "0.61x this rate (profiles/r3_cpu_baseline_config2.json) — run without --cpu-baseline-quick for that number")))


def bf16_record(pipe, dev, inp, a, fp16_forward_out):
    """BASELINE configs[1] names bf16: the same clip with bf16 MFMA operands (identical kernels and speed class), timed
    once, plus the rel-L2 of ONE bf16 denoising forward against the fp16 forward of the same inputs (fp16 is 7e-4 from the
    fp32 reference; bf16's 8-bit mantissa cannot meet the 1e-3 bar) — driver-observed, not gated."""
    dt = torch.bfloat16
    for m in (pipe.denoising_unet, pipe.reference_unet, pipe.vae, pipe.pose_guider, pipe.image_encoder):
        m.to(dtype=dt)
        m.compute_dtype = dt

    def clip():
        emb = pipe.image_encoder(inp["clip_pixels"].to(dt)).image_embeds
        return pipe.run_tensors(inp["ref_image"], inp["bk_images"], inp["pose_images"], emb, inp["latents"], a.ddim_steps, a.guidance)

    clip()
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        clip()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    out = forward_output(pipe, dev, dt, a.size)
    rel = float((out.float() - fp16_forward_out.float()).norm() / fp16_forward_out.float().norm())
    return {"value": a.frames / el, "unit": "frames/s", "ms_per_step": el * 1e3, "steps": n, "forward_rel_l2_vs_fp16": rel}


def fp8_qk_record(pipe, dev, dtype, size, fp16_forward_out, t_fwd_fp16, fam_fp16):
    """BASELINE configs[4] names "fp8 MFMA attention QK" (accuracy reported, not gated): the same
    denoising forward with every spatial attention's Q.K^T on the e4m3 MFMA (mimo_attention_fp8qk, opt-in through
    ops.fp8_qk): rel-L2 of its output against the 16-bit forward of the same inputs, and its time.  The non-scaled fp8 MFMA
    of gfx950 has the f16 rate, and the fp8 variant lives in the generic flash kernel (the d = 40 production kernel keeps
    its integer running max in an f16 k-slot that e4m3 cannot hold), so it is expected to be SLOWER: the record says by how much."""
    from mimo_amd import ops
    with ops.fp8_qk(True):
        out = forward_output(pipe, dev, dtype, size)
        t8, _, _, fam8 = measure_forward(pipe, dev, dtype, size)
    rel = float((out.float() - fp16_forward_out.float()).norm() / fp16_forward_out.float().norm())
    att = lambda f: {"launches": f["attn_kernel"]["launches"], "ms_per_forward": round(f["attn_kernel"]["ms"], 3),
                     "tflops": round(f["attn_kernel"]["flops"] / (f["attn_kernel"]["ms"] * 1e-3) / 1e12, 1)}
    return {"forward_rel_l2_vs_16bit_qk": rel, "forward_ms_fp8_qk": t8 * 1e3, "forward_ms_16bit_qk": t_fwd_fp16 * 1e3,
            "spatial_attention_fp8_qk": att(fam8), "spatial_attention_16bit_qk": att(fam_fp16),
            "note": "opt-in (ops.fp8_qk / mimo_attention_fp8qk): Q.K^T on v_mfma_f32_32x32x16_fp8_fp8, softmax and P.V unchanged"}


def edit_record(pipe, dev, a):
    """BASELINE configs[2] (`--edit`): the run_edit.py path as ONE timed unit — mimo_amd.run_edit.MIMO.run on a synthetic
    template (24 frames of a.size x a.size video / pose / background / occluder, one person blob that fills most of the
    frame): frame selection, ROI-clip segmentation and padding on the host, Pose2VideoPipeline.__call__ (PIL inputs, device
    pre-processing, CLIP, VAE, both UNets), and the per-frame compositing (resize, un-pad, paste, edge mask, occluder) on
    the device, result frames copied to the host — all inside the timed region."""
    import numpy as np
    from mimo_amd.edit import MASK_MODE
    from mimo_amd.run_edit import MIMO, Template
    rs = np.random.RandomState(11)
    S, n = a.size, a.frames
    pose, vid, bk, occ = [], [], [], []
    for i in range(n):
        f = np.zeros((S, S, 3), np.uint8)
        x0 = S // 8 + i
        f[S // 16:S - S // 16, x0:x0 + S // 2] = rs.randint(30, 255, (S - S // 8, S // 2, 3))
        pose.append(f)
        vid.append(rs.randint(0, 255, (S, S, 3), dtype=np.uint8))
        bk.append(rs.randint(0, 255, (S, S, 3), dtype=np.uint8))
        o = np.zeros((S, S, 3), np.uint8)
        o[:, 4 * i:4 * i + S // 10] = 255
        occ.append(o)
    tpl = Template(vid, pose, bk, occ, fps=30)
    mask_list = [rs.rand(256, 256).astype(np.float32) for _ in MASK_MODE]
    ref = rs.randint(0, 256, (S, S * 3 // 4, 3), dtype=np.uint8)
    m = MIMO(pipe, mask_list, width=S, height=S, steps=a.ddim_steps, cfg=a.guidance, seed=42)
    m.run(ref, tpl)  # warm-up
    torch.cuda.synchronize()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        res, _ = m.run(ref, tpl)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    gen = sum(len(c) for c in m.last["context_list"])
    return {"value": len(res) / el, "unit": "output frames/s", "ms_per_clip": el * 1e3, "clips": reps, "output_frames": len(res),
            "generated_frames": gen, "roi_clips": len(m.last["context_list"]),
            "workload": f"BASELINE configs[2]: MIMO.run (run_edit.py:153-306) on a synthetic {S}x{S} template of {n} frames with occluder: "
                        "host template preparation + Pose2VideoPipeline.__call__ + device compositing + frames to the host"}


def forward_output(pipe, dev, dtype, size):
    """One denoising forward (CFG batch of 2 x 24 frames, banks installed) on fixed seeded inputs -> fp32 tokens."""
    from mimo_amd.modules import Ctx, EarlyExit
    from mimo_amd.unet import ReferenceAttentionControl
    h = size // 8
    unet, refu = pipe.denoising_unet, pipe.reference_unet
    g = torch.Generator(device="cpu").manual_seed(7)
    writer = ReferenceAttentionControl(refu, mode="write", do_classifier_free_guidance=True)
    reader = ReferenceAttentionControl(unet, mode="read", do_classifier_free_guidance=True)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)]).to(dev)
    rctx = Ctx(dtype, 1, 1)
    rctx.stop_after = writer.last_block()
    try:
        refu.run_tokens(torch.randn(1, h, h, 8, generator=g).to(dev).to(dtype), 0, ehs[1:], 1, 1, None, rctx)
    except EarlyExit:
        pass
    reader.update(writer)
    x = torch.randn(48, h, h, 8, generator=g).to(dev).to(dtype)
    pose = torch.randn(48, h, h, 320, generator=g).to(dev)
    out = unet.run_tokens(x, 499, ehs, 2, 24, pose).clone()
    reader.clear()
    writer.clear()
    return out


def shard_plan(a, frames, world):
    """--shard-windows: the schedule every rank derives for the clip.  cross_step (default; mimo_amd.pipeline.plan_cross_step):
    slots of (window, step) forwards with one all_gather per slot and no per-step barrier; step_sync (rounds 2-5; plan_items):
    per step, whole windows as b = 2 forwards + single CFG halves as b = 1 forwards.  Costs in batched-window units."""
    if not a.shard_windows:
        return {}
    from mimo_amd.context import get_context_scheduler
    from mimo_amd.pipeline import best_cross_step_plan, cross_step_cost, plan_items, plan_load
    windows = get_context_scheduler("uniform")(0, a.ddim_steps, frames, 24, 1, 4)
    nw = len(windows)
    cfg = a.guidance > 1.0
    load, bound = plan_load(nw, cfg, world)
    sync = {"per_rank_items": [[("window" if len(it) == 2 else ("cond" if it[0][1] else "uncond")) + str(it[0][0]) for it in r]
                               for r in plan_items(nw, cfg, world)],
            "per_rank_cost_per_step": [round(x, 2) for x in load], "busiest_rank_cost": round(max(load) * a.ddim_steps, 2),
            "speedup_bound": round(bound, 2)}
    slots = best_cross_step_plan(windows, a.ddim_steps, world, cfg)
    cost = cross_step_cost(slots)
    cross = {"slots": len(slots), "items_per_slot": sorted({len(sl) for sl in slots}),
             "forwards_per_rank": [sum(1 for sl in slots for it in sl if it[0] == r) for r in range(world)],
             "busiest_rank_cost": round(cost, 2), "speedup_bound": round(nw * a.ddim_steps * (1.0 if cfg else 0.61) / cost, 2),
             "first_slots": [[f"r{r}:w{w}@t{t}" + ("" if len(hv) == 2 or not cfg else ("c" if hv[0] else "u")) for r, w, t, hv in sl] for sl in slots[:3]]}
    return {"shard_plan": {"windows": nw, "units": nw * (2 if cfg else 1), "plan": a.shard_plan, "cross_step": cross, "step_sync": sync,
                           "collective": "all_gather_into_tensor of fp32 [24, h, w, 4] unit predictions per slot (RCCL), "
                                         "exposed_gather_ms = HIP-event time the launch stream waits for the gathers per clip"}}


def emulate_world(a, dev, dtype):
    """--emulate-world W (one GPU): the long-clip mode's per-rank schedule MEASURED instead of modelled.  For every rank k of a
    W-GPU job over a W x 24-frame clip, this GPU runs exactly what rank k would run per clip — CLIP, reference UNet, its chunk
    of the per-frame stages, its work items of every denoising step on two streams, the slot-wise all_gather calls (on a
    world-1 RCCL group: launch path and stream hand-over are real, the xGMI transfer of (W - 1) x 1.5 MB per slot is not),
    the canonical-order accumulation of all units and the DDIM step — and the same clip is timed unsharded on one GPU.
    The busiest rank's time bounds the W-GPU step: speed-up <= single / max_k, before box-to-box spread and link time."""
    import torch.distributed as dist
    from mimo_amd.context import get_context_scheduler
    from mimo_amd.pipeline import best_cross_step_plan, plan_items
    W = a.emulate_world
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29677", rank=0, world_size=1, device_id=dev)
    pipe = build_pipeline(dev, dtype)
    pipe.shard_plan = a.shard_plan
    frames = a.frames * W
    inp = synthetic_inputs(dev, frames, a.size, seed=42)
    cfg = a.guidance > 1.0
    windows = get_context_scheduler("uniform")(0, a.ddim_steps, frames, 24, 1, 4)
    nw = len(windows)
    if a.shard_plan == "cross_step":
        slots = best_cross_step_plan(windows, a.ddim_steps, W, cfg)
        items = [[f"{sum(1 for sl in slots for it in sl if it[0] == k and len(it[3]) == 2)} windows (b = 2) + "
                  f"{sum(1 for sl in slots for it in sl if it[0] == k and len(it[3]) == 1)} halves in {len(slots)} slots"] for k in range(W)]
    else:
        items = [[("window" if len(it) == 2 else ("cond" if it[0][1] else "uncond")) + str(it[0][0]) for it in r] for r in plan_items(nw, cfg, W)]

    def clip():
        emb = pipe.image_encoder(inp["clip_pixels"].to(dtype)).image_embeds
        return pipe.run_tensors(inp["ref_image"], inp["bk_images"], inp["pose_images"], emb, inp["latents"], a.ddim_steps, a.guidance)

    def timed(n=1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            v = clip()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(v).all())
        return (time.perf_counter() - t0) / n * 1e3

    per_rank = []
    pipe.shard_windows = True
    for k in range(W):
        pipe.shard_emulate = (k, W)
        if k == 0:
            clip()  # warm-up (packing, allocator pools)
        pipe.stage_times = {}
        clip()
        st, pipe.stage_times = pipe.stage_times, None
        ms = timed(1)
        per_rank.append({"rank": k, "items": items[k],
                         "ms_per_clip": round(ms, 1), "stage_ms": {n: round(v, 1) for n, v in st.items()}})
    pipe.shard_windows, pipe.shard_emulate = False, None
    clip()
    single = timed(1)
    pipe.batch_invariant = True   # split-K off: the arithmetic the sharded ranks run (bit-identical result)
    clip()
    single_bi = timed(1)
    worst = max(r["ms_per_clip"] for r in per_rank)
    print(json.dumps({"emulate_world": W, "shard_plan": a.shard_plan, "frames": frames, "windows": nw, "size": a.size, "ddim_steps": a.ddim_steps, "dtype": a.dtype,
                      "per_rank": per_rank, "busiest_rank_ms": worst, "single_gpu_ms": round(single, 1),
                      "single_gpu_split_k_off_ms": round(single_bi, 1),
                      "speedup_bound_vs_single": round(single / worst, 2), "speedup_bound_vs_single_split_k_off": round(single_bi / worst, 2),
                      "note": "one GPU measured every rank's share in turn; collectives ran on a world-1 RCCL group (no xGMI time), "
                              "other ranks' results are stand-ins; excludes box-to-box spread", "library_sha256_16": lib_hash()}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emulate-world", type=int, default=0, help="one GPU: measure every rank's share of a W-GPU long-clip job "
                    "(--shard-windows schedule) in turn, and the same clip unsharded; prints the per-rank table as one JSON line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed clips")
    ap.add_argument("--warmup", type=int, default=1, help="untimed clips")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--guidance", type=float, default=3.5)
    ap.add_argument("--edit", action="store_true", help="add the `edit` sub-record: BASELINE configs[2], the whole run_edit.py path timed")
    ap.add_argument("--vae-encode", default="split", choices=["split", "half"],
                    help="precision policy of the VAE encoder (mimo_amd.vae): split = hi + lo operand pairs, 3x the encoder's MFMA work, "
                         "meets the 1e-3 bar in isolation (default, the product's default) | half = plain 16-bit operands")
    ap.add_argument("--unet-precision", default="half", choices=["half", "split"],
                    help="precision policy of both UNets: half = 16-bit MFMA operands + the fused kernels (default, the benchmarked "
                         "path) | split = hi + lo operand pairs, unfused (mimo_amd.precise): the reference-grade mode, ~3x the MFMA work")
    ap.add_argument("--shard-windows", action="store_true")
    ap.add_argument("--edge-split", type=int, default=None,
                    help="mimo_amd.ops.EDGE_SPLIT for this run (default: the shipped 15; 7 = without conv2 / shortcut of the last two "
                         "resnets on split operands: -2 ms per forward, BASELINE configs[0] then measures 1.03e-3)")
    ap.add_argument("--shard-plan", default="cross_step", choices=["cross_step", "step_sync"],
                    help="schedule of the sharded long clip: slots of (window, step) forwards without a per-step barrier (default) | "
                         "the per-step plan of rounds 2-5")
    ap.add_argument("--force-shard", action="store_true",
                    help="with --shard-windows under torchrun at world size 1: take the sharded (RCCL) code path anyway")
    ap.add_argument("--graphs", action="store_true", help="replay the denoising forward as a captured hipGraph "
                    "(measured neutral: the launch queue never runs dry, so eager launches are the default)")
    ap.add_argument("--window-streams", type=int, default=None, help="HIP streams for the independent windows of a step "
                    "(default: the pipeline's setting, 2; 1 = one stream; only clips with > 24 frames have several windows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16 sub-record (one extra clip + one forward)")
    ap.add_argument("--fp8-qk", action="store_true", help="add the fp8 Q.K^T sub-record (BASELINE configs[4]): accuracy and time "
                    "of one denoising forward with the spatial attentions' Q.K^T on the e4m3 MFMA")
    ap.add_argument("--tile-vae", type=int, default=0, help="VAE tiled decode (exact row bands of this many rows; BASELINE configs[4])")
    ap.add_argument("--cpu-baseline-config2", action="store_true", help=argparse.SUPPRESS)  # (the default since round 4)
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="time the CPU baseline on a configs[0]-shaped forward (2 x 8 "
                    "frames at latent 32 x 32, a few seconds) instead of ONE forward at the workload's own configs[1] shape (about a "
                    "minute on 32 threads): quicker, but the CPU sustains ~1.5x the rate there, which flatters it")
    ap.add_argument("--cpu-baseline-only", type=float, default=0.0, help=argparse.SUPPRESS)  # child mode: clip FLOPs
    a = ap.parse_args()
    if a.cpu_baseline_only > 0:
        print(json.dumps(cpu_baseline(a.cpu_baseline_only, a.frames, config2=not a.cpu_baseline_quick)), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: mimo_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment) the RCCL process group is created at
    # ANY world size, so that `torchrun --nproc-per-node 1 bench.py --gpus 1` runs the barrier, the max-over-ranks
    # all_reduce and (with --shard-windows --force-shard) the long-clip exchange on RCCL on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    if a.emulate_world > 1:
        emulate_world(a, dev, dtype)
        return

    from mimo_amd import ops
    pipe = build_pipeline(dev, dtype)
    frames = a.frames * (world if a.shard_windows else 1)
    pipe.shard_windows = a.shard_windows and (world > 1 or (a.force_shard and use_dist))
    pipe.shard_force = bool(a.force_shard)
    pipe.shard_plan = a.shard_plan
    pipe.use_graphs = a.graphs and not pipe.shard_windows
    if a.window_streams is not None:
        pipe.window_streams = a.window_streams
    pipe.vae.encode_precision = a.vae_encode
    pipe.denoising_unet.precision = pipe.reference_unet.precision = a.unet_precision
    if a.edge_split is not None:
        ops.EDGE_SPLIT = a.edge_split
    if a.tile_vae:
        pipe.vae.enable_tiling(a.tile_vae)
    inp = synthetic_inputs(dev, frames, a.size, seed=42 + (0 if a.shard_windows else rank))

    host_video = torch.empty((1, 3, frames, a.size, a.size), dtype=torch.float32, pin_memory=True)

    def clip():
        # the whole __call__ body: CLIP image embedding, VAE encodes, pose guider, reference UNet, the denoising loop,
        # VAE decode (pipeline_pose2vid_long_edit_bkfill_roiclip.py:379-578), and the copy of the video tensor to the
        # host (:125 `.cpu()`; SURVEY 8d: "from inputs on device to video tensor on host")
        emb = pipe.image_encoder(inp["clip_pixels"].to(dtype)).image_embeds
        video = pipe.run_tensors(inp["ref_image"], inp["bk_images"], inp["pose_images"], emb,
                                 inp["latents"], a.ddim_steps, a.guidance)
        host_video.copy_(video, non_blocking=True)
        return video

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        clip()
    barrier()
    if rank == 0 or pipe.shard_windows:  # (the sharded clip contains collectives: every rank has to run it)
        pipe.stage_times = {}
        ops.COUNTER = {"flops": 0, "launches": 0}
        clip()  # one extra untimed clip: per-stage HIP-event marks + executed algorithmic FLOPs / launches
        clip_flops, clip_launches = ops.COUNTER["flops"], ops.COUNTER["launches"]
        ops.COUNTER = None
        stage_ms, pipe.stage_times = pipe.stage_times, None
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        video = clip()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert video.shape == (1, 3, frames, a.size, a.size) and bool(torch.isfinite(video).all())

    if rank == 0:
        total_frames = frames if a.shard_windows else a.frames * world
        ms = elapsed / a.steps * 1e3
        t_fwd, fwd_flops, fwd_launches, fam = measure_forward(pipe, dev, dtype, a.size)
        gk = fam["gemm_kernel"]
        out = {
            "metric": f"denoised frames/sec ({a.size}x{a.size}, {a.frames}f clip, {a.ddim_steps} DDIM steps)", "value": total_frames / (elapsed / a.steps),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong" if a.shard_windows else "weak", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[3]: ONE {a.size}x{a.size} clip of {frames} frames sharded {a.frames} f/GPU over {world} GPUs "
                                    f"({len(range(0, frames, 20)) if frames > 24 else 1} context windows x 2 CFG halves dealt over the ranks, slot-wise "
                                    f"RCCL all_gather per step), " if a.shard_windows else
                                    f"BASELINE configs[{1 if a.size == 512 else 4}]: {a.size}x{a.size}, {a.frames}-frame clip per GPU, independent clip per GPU "
                                    f"(no collective), ") +
                                   f"{a.ddim_steps} DDIM steps, CFG {a.guidance}; CLIP image encoder + reference_unet + pose_guider + VAE enc/dec + "
                                   f"the device-to-host copy of the video inside the timed region",
                       "frames_total": total_frames, "clip_executed_tflop": round(clip_flops / 1e12, 2),
                       "kernel_launches_per_clip": clip_launches,
                       "stage_ms": {k: round(v, 1) for k, v in stage_ms.items()}, "hip_graph": bool(pipe.use_graphs),
                       "vae_tile_rows": a.tile_vae or None, "vae_encode_precision": a.vae_encode, "unet_precision": a.unet_precision,
                       "edge_split": ops.EDGE_SPLIT,
                       "encoder_dedup": "runs of bit-identical input frames are encoded once (the synthetic background is one white frame, as in run_animate.py)",
                       **shard_plan(a, frames, world)},
            # dominant kernel = gemm_kernel (implicit-GEMM convs + linears, ~2/3 of the forward): algorithmic FLOPs of all
            # its launches in one denoising forward / the sum of their HIP-event durations
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel", "achieved": gk["flops"] / (gk["ms"] * 1e-3) / 1e12,
                         "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gk["flops"] / (gk["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS,
                         "traffic": pmc_traffic("gemm_kernel"), "algorithmic_bytes_per_launch": gk["bytes"] / gk["launches"],
                         "launches_per_forward": gk["launches"], "avg_launch_us": gk["ms"] * 1e3 / gk["launches"],
                         "algorithmic_tflop_per_forward": gk["flops"] / 1e12,
                         "attn_kernel": {"achieved": fam["attn_kernel"]["flops"] / (fam["attn_kernel"]["ms"] * 1e-3) / 1e12,
                                         "launches_per_forward": fam["attn_kernel"]["launches"],
                                         "avg_launch_us": fam["attn_kernel"]["ms"] * 1e3 / fam["attn_kernel"]["launches"]},
                         "forward": {"ms": t_fwd * 1e3, "executed_tflop": fwd_flops / 1e12, "launches": fwd_launches,
                                     "achieved": fwd_flops / t_fwd / 1e12, "frac": fwd_flops / t_fwd / 1e12 / PEAK_TFLOPS}},
        }
        if world == 1 and not a.no_bf16 and a.dtype == "fp16":
            try:
                out["bf16"] = bf16_record(pipe, dev, inp, a, forward_output(pipe, dev, dtype, a.size))
            except Exception as e:  # informational sub-record: never lose the headline line
                out["bf16"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if a.edit and world == 1:
            try:
                out["edit"] = edit_record(pipe, dev, a)
            except Exception as e:
                out["edit"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if a.fp8_qk and world == 1:
            try:
                out["fp8_qk"] = fp8_qk_record(pipe, dev, dtype, a.size, forward_output(pipe, dev, dtype, a.size), t_fwd, fam)
            except Exception as e:
                out["fp8_qk"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            import subprocess
            try:  # child process with a hard wall-clock bound: the baseline is informational, never lose the GPU line
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
                                    str(float(clip_flops or fwd_flops * a.ddim_steps)), "--frames", str(a.frames)]
                                   + (["--cpu-baseline-quick"] if a.cpu_baseline_quick else []),
                                   capture_output=True, text=True, timeout=240 if a.cpu_baseline_quick else 600,
                                   env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
                out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": min(os.cpu_count(), 32), "kind": "port",
                                       "sample": f"not measured within {240 if a.cpu_baseline_quick else 600} s: {type(e).__name__}"}
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()
    # under rocprofv3 the interpreter can hang in teardown after the tool has written its output: leave after 60 s
    import threading
    wd = threading.Timer(60.0, os._exit, [0])
    wd.daemon = True
    wd.start()


if __name__ == "__main__":
    main()
