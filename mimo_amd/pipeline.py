"""Pose2VideoPipeline on the MI355X-native models.

Drop-in for src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py::Pose2VideoPipeline (constructor,
`.to()`, `__call__` signature, `.videos` output) as called by run_animate.py:115-123,208-218 and
run_edit.py:116-124,241-251.  `run_tensors()` is the device-resident core (what bench.py times):
CLIP embedding + VAE encode (ref + F background frames) + pose guider + reference UNet (banks, once)
+ {sliding windows x CFG -> denoising UNet -> window average -> guidance -> DDIM} x steps + VAE decode.

Multi-GPU (one process per GPU, torch.distributed over RCCL): the independent work units of a step are
(context window, CFG half) pairs; they are dealt to the ranks heaviest first in a snake (the cond half attends
twice the keys), each rank runs its units as b=1 denoising forwards, and the per-unit predictions travel in
pre-allocated slot buffers: slot k of every rank is all-gathered (async, RCCL's own stream) as soon as the
rank's k-th unit is done, i.e. under the compute of unit k+1; only the last slot's gather is exposed.  Every rank
then replays the window sum in canonical order, so the result is bit-identical to the single-GPU run.
Ranks without a unit (world > units) take part with zero slots.
"""
import math

import torch

from . import ops
from .context import get_context_scheduler
from .unet import ReferenceAttentionControl

VAE_SCALE = 0.18215  # hard-coded by the reference pipeline (:115,431,439)


class Pose2VideoPipelineOutput:
    def __init__(self, videos):
        self.videos = videos


def plan_units(num_windows, cfg, rank, world):
    """Work decomposition of one denoising step.  Units = (window, half) in canonical order
    (w0/uncond, w0/cond, w1/uncond, ...).  Assignment: cond halves (two KV segments: the heavier unit) first, then
    uncond halves, dealt over the ranks in a snake (0..world-1, world-1..0, ...) so that every rank gets the same
    number of units +-1 and heavy / light units alternate per rank.  Returns (all_units, my_units in slot order)."""
    halves = (0, 1) if cfg else (0,)
    units = [(w, h) for w in range(num_windows) for h in halves]
    return units, [u for u, r, _ in assign_units(units, world) if r == rank]


def assign_units(units, world):
    """[(unit, rank, slot)]: the snake deal of plan_units; slot = the unit's index among its rank's units."""
    order = sorted(units, key=lambda u: (-u[1], u[0]))  # cond (half 1) first, window order inside
    out, nslot = [], [0] * world
    for k, u in enumerate(order):
        rnd, pos = divmod(k, world)
        r = pos if rnd % 2 == 0 else world - 1 - pos
        out.append((u, r, nslot[r]))
        nslot[r] += 1
    return out


def _all_gather(send, world, group=None):
    """all_gather of equal-shape tensors.  RCCL (backend "nccl") takes device tensors as they are; the gloo backend
    (CPU tests, and the 2-ranks-on-one-GPU test) has no device all_gather, so device tensors are staged through the host."""
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host = send.cpu()
        recv = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(recv, host, group=group)
        return [r.to(send.device) for r in recv]
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send.contiguous(), group=group)
    return recv


def sharded_frames(fn, x, rank, world, group=None):
    """Per-frame stage (VAE encode / decode, pose guider) over frames [F, ...] with the frames dealt in contiguous
    chunks over the ranks: every rank runs `fn` on its chunk only, one all_gather rebuilds the full result on every
    rank.  Per-frame ops are independent of the batch they run in, so the result equals the unsharded call."""
    F = x.shape[0]
    per = math.ceil(F / world)
    lo, hi = min(F, rank * per), min(F, (rank + 1) * per)
    # a rank past the end (F < world * per) still joins the collective with a dummy frame
    mine = fn(x[lo:hi] if hi > lo else x[:1])
    send = torch.zeros((per,) + tuple(mine.shape[1:]), device=mine.device, dtype=mine.dtype)
    if hi > lo:
        send[:hi - lo] = mine
    return torch.cat(_all_gather(send, world, group), 0)[:F].contiguous()


class UnitExchange:
    """Pre-allocated exchange of per-unit predictions (fp32 [Fw, h, w, C] each).

    Rank r owns ceil/floor(len(units)/world) slots.  `put(slot, pred)` copies the prediction into the rank's send
    slot and immediately starts the all-gather of THAT slot index over all ranks (async_op: RCCL runs it on its own
    stream, ordered after the producing kernels by an event), so the gather of slot k overlaps the compute of the
    rank's unit k+1.  `finish()` starts the gathers of slots this rank does not fill (it sends zeros there), waits
    for all of them and returns {unit: tensor}.  Buffers are allocated once per (shape, world) and reused every step."""

    def __init__(self, units, rank, world, shape, device, group=None):
        self.units, self.rank, self.world, self.group = units, rank, world, group
        self.assign = assign_units(units, world)
        self.nslots = math.ceil(len(units) / world) if units else 0
        self.send = torch.zeros((self.nslots,) + tuple(shape), device=device, dtype=torch.float32)
        self.recv = torch.zeros((self.nslots, world) + tuple(shape), device=device, dtype=torch.float32)
        self.my_slots = sum(1 for _, r, _ in self.assign if r == rank)
        self._host = None
        self.begin()

    def begin(self):
        self.works, self.started = [], 0

    def _gather_slot(self, k):
        import torch.distributed as dist
        if self.send.is_cuda and dist.get_backend(self.group) == "gloo":  # no device collectives in gloo: host staging
            host = self.send[k].reshape(-1).cpu()
            out = torch.empty((self.world * host.numel(),), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host, group=self.group)
            self.recv[k].copy_(out.view(self.recv[k].shape))
            return
        # flat views: the concatenated form of all_gather_into_tensor is accepted by every backend
        self.works.append(dist.all_gather_into_tensor(self.recv[k].view(-1), self.send[k].view(-1), group=self.group,
                                                      async_op=True))

    def put(self, pred):
        k = self.started
        self.send[k].copy_(pred)
        self._gather_slot(k)
        self.started += 1

    def finish(self):
        while self.started < self.nslots:  # slots this rank has no unit for carry zeros
            self.send[self.started].zero_()
            self._gather_slot(self.started)
            self.started += 1
        for w in self.works:
            w.wait()
        out = {u: self.recv[slot, r] for u, r, slot in self.assign}
        self.begin()
        return out


def exchange_predictions(my_preds, units, rank, world, group=None, shape=None, device=None):
    """One-shot form of UnitExchange: {unit: tensor} of this rank's units -> {unit: tensor} for ALL units.  A rank that
    owns no unit (world > len(units)) must pass `shape` / `device` (known on the host: [Fw, h, w, C] fp32)."""
    if my_preds:
        proto = next(iter(my_preds.values()))
        shape, device = tuple(proto.shape), proto.device
    elif shape is None:
        raise ValueError("a rank without units must be given the unit shape")
    ex = UnitExchange(units, rank, world, shape, device or "cpu", group)
    for u, r, _ in ex.assign:  # slot order of this rank
        if r == rank:
            ex.put(my_preds[u])
    return ex.finish()


class GraphedDenoiser:
    """The denoising-UNet forward for one (batch, window, latent) shape captured as ONE hipGraph (torch.cuda.CUDAGraph
    over the C-ABI launches, which all go to the capture stream): ~430 kernel launches and the host-side dispatch
    collapse into a single replay.  Inputs live in static buffers; the bank K/V tensors are persistent per block."""

    def __init__(self, unet, b, F, h, w, c0, dtype):
        dev = unet.device
        self.unet, self.b, self.F = unet, b, F
        self.x = torch.zeros((b * F, h, w, 8), device=dev, dtype=dtype)
        self.pose = torch.zeros((b * F, h, w, c0), device=dev, dtype=torch.float32)
        self.t = torch.zeros((1,), device=dev, dtype=torch.float32)
        self.ehs = torch.zeros((b, 1, unet.cross_dim), device=dev, dtype=torch.float32)
        self.graph = None
        self.out = None

    def capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture: weight packing, lazy library load
            self.unet.run_tokens(self.x, self.t, self.ehs, self.b, self.F, self.pose)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        saved, ops.COUNTER = ops.COUNTER, {"flops": 0, "launches": 0}
        with torch.cuda.graph(self.graph):
            self.out = self.unet.run_tokens(self.x, self.t, self.ehs, self.b, self.F, self.pose)
        self.work, ops.COUNTER = ops.COUNTER, saved  # algorithmic FLOPs / launches replayed by every graph launch

    def fingerprint(self):
        """Everything a captured graph bakes in by ADDRESS or by VALUE besides the static input buffers: the bank K/V
        buffers of the spatial blocks, the packed-weight generation (HipModule.invalidate drops packed buffers on
        .to() / load_state_dict) and the split-K setting.  A replay with a different fingerprint would read freed or
        stale memory, so the graph is re-captured instead."""
        from .modules import pack_epoch
        banks = tuple(0 if b.bank_kv is None else b.bank_kv.data_ptr() for b in self.unet.spatial_blocks())
        return (banks, pack_epoch(), ops.split_k_enabled())

    def __call__(self, x, t, ehs, pose):
        self.x.copy_(x)
        self.pose.copy_(pose)
        self.t.fill_(float(t))
        self.ehs.copy_(ehs)
        if self.graph is not None and self.fp != self.fingerprint():
            self.graph = None  # stale addresses: drop and re-capture
        if self.graph is None:
            self.capture()
            self.fp = self.fingerprint()
        self.graph.replay()
        if ops.COUNTER is not None:
            ops.COUNTER["flops"] += self.work["flops"]
            ops.COUNTER["launches"] += self.work["launches"]
        return self.out


class Pose2VideoPipeline:
    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, pose_guider, scheduler,
                 image_proj_model=None, tokenizer=None, text_encoder=None):
        self.vae, self.image_encoder = vae, image_encoder
        self.reference_unet, self.denoising_unet, self.pose_guider = reference_unet, denoising_unet, pose_guider
        self.scheduler = scheduler
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.vae_batch = 8  # frames per VAE launch group (bounds activation memory; results are per-image)
        self.dist_group = None
        self.shard_windows = False  # True: deal (window, CFG half) units over the torch.distributed ranks
        self.batch_invariant = False  # True: bit-identical to the sharded run of the same clip (split-K off)
        self.use_graphs = False     # True: replay the denoising forward as a captured hipGraph (single-GPU path)
        self.window_streams = 2     # > 1: the independent windows of one step run on this many HIP streams (single-GPU path)
        self._side_streams = {}
        self.stage_times = None     # dict -> accumulates per-stage milliseconds (HIP events) of run_tensors
        self._graphs = {}

    def to(self, device=None, dtype=None):
        for m in (self.vae, self.image_encoder, self.reference_unet, self.denoising_unet, self.pose_guider):
            if isinstance(m, torch.nn.Module):
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        return self.denoising_unet.device

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    # ------------------------------------------------------------------------------------------
    def _encode_frames(self, images):
        """images [n,3,H,W] in [-1,1] (device), or half tokens [n,H,W,8] from image.vae_preprocess
        -> fp32 latent tokens [n,h,w,4] * 0.18215 (pipeline :427-443)."""
        dt = self.vae.compute_dtype
        outs = []
        tokens_in = images.shape[-1] == 8 and images.dtype == dt
        nb = min(self.vae_batch, self.vae.max_images(*(images.shape[1:3] if tokens_in else images.shape[-2:])))
        for i in range(0, images.shape[0], nb):
            if tokens_in:
                tok = images[i:i + nb].contiguous()
            else:
                tok = ops.ncfhw_to_tokens(images[i:i + nb].float().contiguous()[:, :, None], dt, cpad=8)
            outs.append(self.vae.encode_tokens(tok))
        return torch.cat(outs) * VAE_SCALE

    def _decode_frames(self, latents):
        """latents fp32 [1,4,F,h,w] -> video fp32 [1,3,F,H,W] in [0,1] (decode_latents, pipeline :113-126)."""
        dt = self.vae.compute_dtype
        _, C, F, h, w = latents.shape
        z = ops.ncfhw_to_tokens((latents * (1.0 / VAE_SCALE)).contiguous(), dt, cpad=8)  # [F,h,w,8]
        frames = []
        nb = min(self.vae_batch, self.vae.max_images(8 * h, 8 * w))
        for i in range(0, F, nb):
            y = self.vae.decode_tokens(z[i:i + nb].contiguous())
            frames.append(ops.tokens_to_image(y, y.shape[0], y.shape[1], y.shape[2]))
        return torch.cat(frames).permute(1, 0, 2, 3)[None]

    @torch.no_grad()
    def run_tensors(self, *args, **kwargs):
        """Device-resident core.  ref_image [1,3,H,W] in [-1,1]; bk_images [F,3,H,W] in [-1,1]; pose_images
        [F,3,H,W] in [0,1]; clip_embeds [1,768]; latents fp32 [1,4,F,h,w] (the injected initial noise).
        (Signature: see _run_tensors.)  Split-K is a per-call flag of the C-ABI: it is switched off for THIS run only
        when the summation order must not depend on how the clip is cut into batches."""
        world = 1
        if self.shard_windows and torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(self.dist_group)
        with ops.split_k(not (world > 1 or self.batch_invariant)):
            return self._run_tensors(*args, **kwargs)

    def _run_tensors(self, ref_image, bk_images, pose_images, clip_embeds, latents, num_inference_steps,
                     guidance_scale, context_schedule="uniform", context_frames=24, context_stride=1,
                     context_overlap=4, callback=None, return_latents=False, decode=True, trajectory=None):
        dev = self.device
        unet, sched = self.denoising_unet, self.scheduler
        dt = unet.compute_dtype
        cfg = guidance_scale > 1.0
        rank, world = 0, 1
        if self.shard_windows and torch.distributed.is_available() and torch.distributed.is_initialized():
            import torch.distributed as dist
            rank, world = dist.get_rank(self.dist_group), dist.get_world_size(self.dist_group)
        sched.set_timesteps(num_inference_steps)
        latents = latents.to(device=dev, dtype=torch.float32).contiguous().clone()
        _, C, F, h, w = latents.shape

        ehs_c = clip_embeds.to(dev).float().reshape(1, 1, -1)
        ehs = torch.cat([torch.zeros_like(ehs_c), ehs_c], 0) if cfg else ehs_c

        # VAE encode: reference image + F background frames; pose guider
        marks = []  # (stage name, event) pairs when self.stage_times is a dict (bench instrumentation)

        def mark(name):
            if self.stage_times is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))

        mark("start")
        ref_lat = self._encode_frames(ref_image.to(dev))                       # [1,h,w,4]

        def pose_fn(frames):
            if frames.shape[-1] == 8 and frames.dtype == self.pose_guider.compute_dtype:
                tok = frames  # already half tokens (image.vae_preprocess)
            else:
                tok = ops.ncfhw_to_tokens(frames.float().contiguous()[:, :, None], self.pose_guider.compute_dtype, cpad=8)
            return torch.cat([self.pose_guider.run_tokens(tok[i:i + self.vae_batch].contiguous())
                              for i in range(0, tok.shape[0], self.vae_batch)])

        # reference UNet at t = 0 (pipeline :480-490): only the cond element's banks are ever read, and
        # everything after the last bank write is dead code -> run b = 1 with early exit.  It is a chain of small (4096-token)
        # launches that depends only on ref_lat: on the single-GPU path it runs on a side stream under the VAE encode of the
        # background frames and the pose guider (which fill the chip with large launches).
        import contextlib
        from .modules import Ctx, EarlyExit
        writer = ReferenceAttentionControl(self.reference_unet, mode="write", do_classifier_free_guidance=cfg)
        reader = ReferenceAttentionControl(unet, mode="read", do_classifier_free_guidance=cfg)
        main = torch.cuda.current_stream(dev)
        side = None
        if world == 1 and self.window_streams > 1:
            side = self._side_streams.setdefault(dev, [torch.cuda.Stream(dev) for _ in range(self.window_streams)])[0]
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()), ops.workspace_slot(1 if side is not None else 0):
            ref_tok = torch.zeros((1, h, w, 8), device=dev, dtype=self.reference_unet.compute_dtype)
            ref_tok[..., :C] = ref_lat.to(ref_tok.dtype)
            rctx = Ctx(self.reference_unet.compute_dtype, 1, 1)
            rctx.stop_after = writer.last_block()
            try:
                self.reference_unet.run_tokens(ref_tok, 0, ehs_c, 1, 1, None, rctx)
            except EarlyExit:
                pass

        if world > 1:  # one long clip over the ranks: the per-frame stages are sharded too
            bk_tok = sharded_frames(self._encode_frames, bk_images.to(dev), rank, world, self.dist_group).to(dt)
            pose_tok = sharded_frames(pose_fn, pose_images.to(dev), rank, world, self.dist_group)
        else:
            bk_tok = self._encode_frames(bk_images.to(dev)).to(dt)             # [F,h,w,4]
            pose_tok = pose_fn(pose_images.to(dev))                            # fp32 [F,h,w,C0]

        if side is not None:
            main.wait_stream(side)
            for blk in self.reference_unet.spatial_blocks():  # the banked tensors were allocated on the side stream
                for bt in blk.bank:
                    bt.record_stream(main)
        reader.update(writer)
        mark("vae_encode+pose_guider+reference_unet")

        windows = get_context_scheduler(context_schedule)(0, num_inference_steps, F, context_frames, context_stride,
                                                          context_overlap)
        win_idx = [torch.tensor(c, dtype=torch.int32, device=dev) for c in windows]
        win_bk = [bk_tok[c.long()] for c in win_idx]
        win_pose = [pose_tok[c.long()] for c in win_idx]
        units, my_units = plan_units(len(windows), cfg, rank, world)
        exch = None
        if world > 1:  # every window has the same frame count; the UNet's output head is padded to 4 channels
            cpad = (unet.out_channels + 3) // 4 * 4
            exch = UnitExchange(units, rank, world, (len(windows[0]), h, w, cpad), dev, self.dist_group)
        acc = torch.empty((2 if cfg else 1, C, F, h, w), device=dev, dtype=torch.float32)
        counter = torch.empty((F,), device=dev, dtype=torch.float32)

        for step, t in enumerate(sched.timesteps.tolist()):
            acc.zero_()
            counter.zero_()
            preds = {}
            if world == 1 and len(win_idx) > 1 and self.window_streams > 1 and not self.use_graphs:
                # The windows of one step are independent forwards: on two streams their kernels fill each other's idle
                # CUs (tail rounds of small levels, HBM-bound linears beside MFMA-bound convolutions): -4.7 % per step at
                # two windows (profiles/r2_two_stream_forward.txt).  Accumulation stays in canonical window order.
                main = torch.cuda.current_stream(dev)
                streams = self._side_streams.setdefault(dev, [torch.cuda.Stream(dev) for _ in range(self.window_streams)])
                for s_ in streams:
                    s_.wait_stream(main)
                rep = 2 if cfg else 1
                wpred = []
                for wi, idx in enumerate(win_idx):
                    slot = wi % len(streams)
                    with torch.cuda.stream(streams[slot]), ops.workspace_slot(1 + slot):
                        lat_tok = ops.ncfhw_to_tokens(latents, dt, frame_idx=idx)
                        x = torch.cat([lat_tok, win_bk[wi]], dim=-1)
                        wpred.append(unet.run_tokens(x.repeat(rep, 1, 1, 1), t, ehs, rep, idx.numel(),
                                                     win_pose[wi].repeat(rep, 1, 1, 1)))
                for s_ in streams:
                    main.wait_stream(s_)
                for pred, idx in zip(wpred, win_idx):
                    pred.record_stream(main)
                    ops.window_accumulate(pred, idx, acc, counter)
            elif world == 1:
                for wi, idx in enumerate(win_idx):
                    lat_tok = ops.ncfhw_to_tokens(latents, dt, frame_idx=idx)                   # [Fw,h,w,4]
                    x = torch.cat([lat_tok, win_bk[wi]], dim=-1)
                    rep = 2 if cfg else 1
                    if self.use_graphs:
                        key = (rep, idx.numel(), h, w, dt)
                        if key not in self._graphs:
                            self._graphs[key] = GraphedDenoiser(unet, rep, idx.numel(), h, w, pose_tok.shape[-1], dt)
                        pred = self._graphs[key](x.repeat(rep, 1, 1, 1), t, ehs, win_pose[wi].repeat(rep, 1, 1, 1))
                    else:
                        pred = unet.run_tokens(x.repeat(rep, 1, 1, 1), t, ehs, rep, idx.numel(),
                                               win_pose[wi].repeat(rep, 1, 1, 1))
                    ops.window_accumulate(pred, idx, acc, counter)
            else:
                # a rank's units are independent b = 1 forwards: with two or more they run on alternating HIP streams
                # (two half-batch forwards overlap 15 % better than back to back, profiles/r2_two_stream_forward.txt);
                # each prediction is handed to the exchange, in unit order, as soon as its stream is done
                conc = len(my_units) > 1 and self.window_streams > 1
                main = torch.cuda.current_stream(dev)
                streams = self._side_streams.setdefault(dev, [torch.cuda.Stream(dev) for _ in range(self.window_streams)]) if conc else [main]
                if conc:
                    for s_ in streams:
                        s_.wait_stream(main)
                pending = []
                for k, (wi, half) in enumerate(my_units):
                    idx = win_idx[wi]
                    slot = k % len(streams)
                    with torch.cuda.stream(streams[slot]), ops.workspace_slot(1 + slot if conc else 0):
                        lat_tok = ops.ncfhw_to_tokens(latents, dt, frame_idx=idx)
                        x = torch.cat([lat_tok, win_bk[wi]], dim=-1)
                        e = ehs[half:half + 1] if cfg else ehs
                        pred = self._run_unit(unet, x, t, e, idx.numel(), win_pose[wi], cond=(half == 1 or not cfg))
                    pending.append((pred, streams[slot]))
                    # the oldest pending unit is handed over once every stream has a younger unit queued behind it
                    if len(pending) == len(streams):
                        p0, s0 = pending.pop(0)
                        if conc:
                            main.wait_stream(s0)
                            p0.record_stream(main)
                        exch.put(p0)  # the gather of this unit's slot starts now and runs under the following units
                for p0, s0 in pending:
                    if conc:
                        main.wait_stream(s0)
                        p0.record_stream(main)
                    exch.put(p0)
                allp = exch.finish()
                for wi, idx in enumerate(win_idx):
                    halves = [allp[(wi, hf)] for hf in ((0, 1) if cfg else (0,))]
                    ops.window_accumulate(torch.cat(halves, 0).contiguous(), idx, acc, counter)
            ops.cfg_ddim_step(acc, counter, latents, cfg, guidance_scale, *sched.coefficients(t))
            if trajectory is not None:
                trajectory.append(latents.clone())
            if callback is not None:
                callback(step, t, latents)
        reader.clear()
        writer.clear()
        mark("denoising_loop")
        if not decode:
            return latents
        if world > 1:
            fr = sharded_frames(lambda z: self._decode_frames(z.permute(1, 0, 2, 3)[None].contiguous())[0].permute(1, 0, 2, 3),
                                latents[0].permute(1, 0, 2, 3).contiguous(), rank, world, self.dist_group)
            video = fr.permute(1, 0, 2, 3)[None].contiguous()
        else:
            video = self._decode_frames(latents)
        mark("vae_decode")
        if self.stage_times is not None:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                self.stage_times[n1] = self.stage_times.get(n1, 0.0) + e0.elapsed_time(e1)
        return (video, latents) if return_latents else video

    def _run_unit(self, unet, x, t, ehs1, Fw, pose, cond):
        """One (window, CFG half) unit as a b = 1 forward.  The uncond half must not read the bank."""
        blocks = unet.spatial_blocks()
        saved = None
        if not cond:
            saved = [b.bank_kv for b in blocks]
            for b in blocks:
                b.bank_kv = None
        try:
            return unet.run_tokens(x, t, ehs1, 1, Fw, pose)
        finally:
            if saved is not None:
                for b, kv in zip(blocks, saved):
                    b.bank_kv = kv

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, ref_image, pose_images, vid_bk_images, width, height, video_length, num_inference_steps,
                 guidance_scale, num_images_per_prompt=1, eta=0.0, generator=None, output_type="tensor",
                 return_dict=True, callback=None, callback_steps=1, context_schedule="uniform", context_frames=24,
                 context_stride=1, context_overlap=4, context_batch_size=1, interpolation_factor=1, **kwargs):
        if eta != 0.0 or context_batch_size != 1 or interpolation_factor != 1:
            raise NotImplementedError("eta=0, context_batch_size=1, interpolation_factor=1 (the reference defaults) only")
        dev = self.device
        from . import image as IM
        # CLIP image embedding (pipeline :379-384): `ref_image.resize((224, 224))` + CLIPImageProcessor run on the device
        # (PIL-exact bicubic resample + rescale / normalise); any module returning `.image_embeds` works as the encoder.
        clip_image = IM.clip_preprocess(ref_image, dev)
        enc_dtype = getattr(self.image_encoder, "dtype", torch.float32)
        clip_embeds = self.image_encoder(clip_image.to(dtype=enc_dtype)).image_embeds
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        # prepare_latents (:149-183): CPU generator draws on CPU in the embedding dtype, then moves
        shape = (1, 4, video_length, h, w)
        gdev = generator.device.type if generator is not None else "cpu"
        if gdev == "cpu":
            latents = torch.randn(shape, generator=generator, device="cpu", dtype=clip_embeds.dtype).to(dev)
        else:
            latents = torch.randn(shape, generator=generator, device=dev, dtype=clip_embeds.dtype)
        # VaeImageProcessor.preprocess (:424-457) on the device: the host only decodes the PIL images to raw bytes;
        # LANCZOS resize, /255, 2x-1 and the token layout are kernels (image.vae_preprocess)
        vdt = self.vae.compute_dtype
        ref_t = IM.vae_preprocess([ref_image], height, width, True, vdt, dev)
        bk_t = IM.vae_preprocess(list(vid_bk_images), height, width, True, vdt, dev)
        pose_t = IM.vae_preprocess(list(pose_images), height, width, False, self.pose_guider.compute_dtype, dev)
        cb = None
        if callback is not None:
            cb = lambda i, t, lat: callback(i, t, lat) if i % callback_steps == 0 else None
        video = self.run_tensors(ref_t, bk_t, pose_t, clip_embeds, latents.float(), num_inference_steps, guidance_scale,
                                 context_schedule, context_frames, context_stride, context_overlap, callback=cb)
        images = video.cpu().float()
        if output_type != "tensor":
            images = images.numpy()
        return Pose2VideoPipelineOutput(videos=images) if return_dict else images
