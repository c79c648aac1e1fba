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


# src/pipelines/utils.py: the interpolation used by interpolate_latents is a module-level choice that nothing in the
# reference ever makes (set_tensor_interpolation_method is never called: interpolation_factor >= 2 raises TypeError there)
tensor_interpolation = None


def get_tensor_interpolation_method():
    return tensor_interpolation


def set_tensor_interpolation_method(is_slerp):
    global tensor_interpolation
    tensor_interpolation = slerp if is_slerp else linear


def linear(v1, v2, t):
    return (1.0 - t) * v1 + t * v2


def slerp(v0, v1, t, DOT_THRESHOLD=0.9995):
    u0, u1 = v0 / v0.norm(), v1 / v1.norm()
    dot = (u0 * u1).sum()
    if dot.abs() > DOT_THRESHOLD:
        return (1.0 - t) * v0 + t * v1
    omega = dot.acos()
    return (((1.0 - t) * omega).sin() * v0 + (t * omega).sin() * v1) / omega.sin()


def interpolate_latents(latents, interpolation_factor):
    """pipeline_pose2vid_long_edit_bkfill_roiclip.py:293-336: (F - 1) * factor + 1 frames, the new ones interpolated between
    neighbours with the method chosen by set_tensor_interpolation_method.  Off the hot path (factor 1 = the default =
    identity); a handful of elementwise torch ops per inserted frame, on the device the latents live on."""
    if interpolation_factor < 2:
        return latents
    fn = get_tensor_interpolation_method()
    if fn is None:
        raise TypeError("interpolation_factor >= 2 needs set_tensor_interpolation_method(is_slerp) first "
                        "(src/pipelines/utils.py; the reference fails with 'NoneType is not callable' here)")
    F = latents.shape[2]
    rate = [i / interpolation_factor for i in range(interpolation_factor)][1:]
    frames = []
    for i0 in range(F - 1):
        v0, v1 = latents[:, :, i0], latents[:, :, i0 + 1]
        frames.append(v0)
        frames += [fn(v0, v1, f) for f in rate]
    frames.append(latents[:, :, F - 1])
    return torch.stack(frames, dim=2).contiguous()


def fill_window_tokens(latents, idx, out, C, dt):
    """latents fp32 [1, C, F, h, w], frames idx -> token rows out[..., :C] ([len(idx), h, w, >= C]): the layout kernel for 16-bit
    tokens; a torch layout copy for fp32 tokens (the UNets' split precision policy takes its input unrounded)."""
    if out.dtype == torch.float32:
        out[..., :C] = latents[0][:, idx.long()].permute(1, 2, 3, 0)
    else:
        ops.ncfhw_to_tokens(latents, dt, frame_idx=idx, cpad=C, out=out)
    return out


class Pose2VideoPipelineOutput:
    def __init__(self, videos):
        self.videos = videos


# Relative cost of the three kinds of work item of one denoising step, in units of ONE batched window forward (b = 2:
# uncond + cond rows in one launch sequence).  Measured on one MI355X (profiles/r2_two_stream_forward.txt): batched
# 83.1 ms; the same window as two b = 1 forwards 108 ms back to back (half-size launches quantise worse), of which the
# cond half — it attends [self || bank], 1.2518 vs 1.1037 TFLOP per frame — is the heavier one.
ITEM_COST = {"window": 1.0, "cond": 0.69, "uncond": 0.61}


def _item_cost(item):
    if len(item) == 2:
        return ITEM_COST["window"]
    return ITEM_COST["cond"] if item[0][1] == 1 else ITEM_COST["uncond"]


def plan_items(num_windows, cfg, world):
    """Work items of one denoising step, per rank: [[item, ...] for each rank]; an item is a tuple of the (window, half)
    units ONE forward produces: ((w, 0), (w, 1)) = the whole window batched (b = 2), ((w, h),) = one CFG half (b = 1).

    Whole windows are dealt first — a batched window costs 1.0, its two halves 1.30 — and only the last k windows are cut
    into halves to level the ranks; k and the deal (longest item first onto the least loaded rank, ties to the lowest
    rank) minimise the busiest rank's cost.  BASELINE configs[3] (10 windows, 8 ranks): 8 whole windows + 4 halves ->
    four ranks carry 1.69 / 1.61, four carry 1.0: 10 / 1.69 = 5.9x (6.07x at the measured 54 ms per half) against 5.0x
    for whole windows only and 5.2x for 20 half units.  Without CFG every window is one b = 1 item."""
    if not cfg:
        items = [((w, 0),) for w in range(num_windows)]
        return _deal(items, world)
    best = None
    for k in range(num_windows + 1):
        items = [((w, 0), (w, 1)) for w in range(num_windows - k)]
        items += [((w, h),) for w in range(num_windows - k, num_windows) for h in (1, 0)]
        ranks = _deal(items, world)
        load = max(sum(_item_cost(i) for i in r) for r in ranks)
        if best is None or load < best[0] - 1e-9:
            best = (load, ranks)
    return best[1]


def _deal(items, world):
    order = sorted(items, key=lambda i: (-_item_cost(i), i[0][0], -i[0][1]))
    ranks, load = [[] for _ in range(world)], [0.0] * world
    for it in order:
        r = min(range(world), key=lambda q: (round(load[q], 9), q))
        ranks[r].append(it)
        load[r] += _item_cost(it)
    return ranks


def plan_load(num_windows, cfg, world):
    """(per-rank cost in window-forward units, speed-up bound = num_windows (x 0.61 without CFG) / busiest rank)."""
    ranks = plan_items(num_windows, cfg, world)
    load = [sum(_item_cost(i) for i in r) for r in ranks]
    one = num_windows * (ITEM_COST["window"] if cfg else ITEM_COST["uncond"])
    return load, one / max(load) if max(load) > 0 else 1.0


def assign_units(units, world, cfg=None):
    """[(unit, rank, slot)] of plan_items: slot = the unit's index among its rank's units (item order; the two halves
    of a whole window take consecutive slots, uncond first)."""
    if cfg is None:
        cfg = any(h == 1 for _, h in units)
    nw = 1 + max(w for w, _ in units) if units else 0
    out = []
    for r, items in enumerate(plan_items(nw, cfg, world)):
        slot = 0
        for it in items:
            for u in it:
                out.append((u, r, slot))
                slot += 1
    return out


def plan_units(num_windows, cfg, rank, world):
    """Work decomposition of one denoising step.  Units = (window, half) in canonical order (w0/uncond, w0/cond,
    w1/uncond, ...); returns (all_units, this rank's units in slot order).  See plan_items for the deal."""
    halves = (0, 1) if cfg else (0,)
    units = [(w, h) for w in range(num_windows) for h in halves]
    return units, [u for u, r, _ in assign_units(units, world, cfg) if r == rank]


# ---------------------------------------------------------------------------------------------------------------------
# Cross-step schedule of the sharded long clip (round 6).  Window w at step t + 1 needs the step-t predictions of the windows
# that share a frame with it — w itself and its ring neighbours (4-frame overlaps, src/pipelines/context.py:15-42) — and of
# nothing else: the reference's per-step loop (pipeline_pose2vid_long_edit_bkfill_roiclip.py:505-553) has no global barrier in
# its arithmetic.  The (window, step) forwards are therefore laid into SLOTS of at most `world` items by a static list schedule;
# every slot ends with ONE async all_gather of the ranks' predictions, after which every rank advances (window sum in canonical
# order -> guidance -> DDIM, the single-GPU arithmetic per element) exactly the frames all of whose covering windows have
# delivered.  BASELINE configs[3] (10 windows x 20 steps on 8 ranks): 200 whole-window (b = 2) forwards in 25 full slots —
# every rank runs 25 forwards — where the step-synchronous plan (plan_items) keeps four ranks busy for 1.61 window-forwards
# per step and idles the others: 32.2 against 25.0 window-forwards on the busiest rank.
# ---------------------------------------------------------------------------------------------------------------------


def window_neighbours(windows):
    """nb[w] = the windows (w included) that share at least one frame with window w, ascending."""
    sets = [set(c) for c in windows]
    return [[v for v in range(len(windows)) if sets[w] & sets[v]] for w in range(len(windows))]


def interleaved_order(n):
    """0, n-1, 1, n-2, ...: on a ring of n windows every window's neighbours sit within two positions of it, so the items
    of step t + 1 become ready in the order those of step t are issued."""
    out, lo, hi = [], 0, n - 1
    while lo <= hi:
        out.append(lo)
        if hi != lo:
            out.append(hi)
        lo, hi = lo + 1, hi - 1
    return out


def plan_cross_step(windows, steps, world, cfg=True, order=None):
    """Static schedule of the (window, step) forwards: a list of slots, each a list of (rank, w, t, halves) with halves = (0, 1)
    for a whole window (one b = 2 forward when cfg, else (0,)) or (h,) for one CFG half (a b = 1 forward).  An item enters a
    slot only if every step-(t - 1) item of its neighbour windows sits in an EARLIER slot (their predictions are gathered at
    the end of that slot).  Greedy by (step, position in `order`); a slot whose ready windows fill at most half the ranks runs
    them as CFG halves on two ranks each (a half costs 0.69 of a batched window).  Deterministic: every rank computes the
    same plan."""
    nw = len(windows)
    nb = window_neighbours(windows)
    pos = {w: i for i, w in enumerate(order if order is not None else interleaved_order(nw))}
    pending = sorted(((t, pos[w], w) for t in range(steps) for w in range(nw)))
    done = {}
    slots = []
    while pending:
        s = len(slots)
        ready = []
        for it in pending:
            t, _, w = it
            if t == 0 or all(done.get((v, t - 1), s) < s for v in nb[w]):
                ready.append(it)
                if len(ready) == world:
                    break
        assert ready, "dependency cycle in the window schedule"
        slot = []
        if cfg and 2 * len(ready) <= world:
            for j, (t, _, w) in enumerate(ready):
                slot += [(2 * j, w, t, (1,)), (2 * j + 1, w, t, (0,))]   # the cond half (it attends the bank too) first
        else:
            for j, (t, _, w) in enumerate(ready):
                slot.append((j, w, t, (0, 1) if cfg else (0,)))
        for t, _, w in ready:
            done[(w, t)] = s
        taken = set(ready)
        pending = [it for it in pending if it not in taken]
        slots.append(slot)
    return slots


def cross_step_cost(slots):
    """Busiest-rank cost of a plan in batched-window units: every slot lasts as long as its most expensive item."""
    def c(halves):
        return ITEM_COST["window"] if len(halves) == 2 else ITEM_COST["cond"] if halves[0] == 1 else ITEM_COST["uncond"]
    return sum(max(c(h) for _, _, _, h in slot) for slot in slots)


def best_cross_step_plan(windows, steps, world, cfg=True):
    """The cheaper of the interleaved and the natural window order (cross_step_cost; ties to the interleaved one)."""
    a = plan_cross_step(windows, steps, world, cfg)
    b = plan_cross_step(windows, steps, world, cfg, order=list(range(len(windows))))
    return a if cross_step_cost(a) <= cross_step_cost(b) + 1e-9 else b


def frame_segments(windows, num_frames):
    """[(cover, frames)]: the frames grouped by the set of windows that contain them (cover = ascending window tuple), in
    order of first frame.  A segment advances from step t to t + 1 when every window of its cover has delivered step t."""
    cover = [[] for _ in range(num_frames)]
    for w, c in enumerate(windows):
        for f in c:
            cover[f].append(w)
    segs = {}
    for f in range(num_frames):
        segs.setdefault(tuple(cover[f]), []).append(f)
    return sorted(segs.items(), key=lambda kv: kv[1][0])


def cross_step_lifetime(slots, windows, num_frames):
    """How many slots a gathered prediction must stay readable: max over items of (slot in which the last segment it feeds
    becomes ready) - (its own slot) + 1."""
    done = {(w, t): s for s, slot in enumerate(slots) for _, w, t, _ in slot}
    life = 1
    for cover, _ in frame_segments(windows, num_frames):
        for t in {t for _, t in done}:
            last = max(done[(w, t)] for w in cover)
            life = max(life, 1 + max(last - done[(w, t)] for w in cover))
    return life


class SlotExchange:
    """Per-slot exchange of the cross-step schedule: every rank contributes up to two unit predictions (fp32 [Fw, h, w, C]:
    uncond + cond of a whole window, or one CFG half) per slot; slot s of all ranks is all-gathered asynchronously (RCCL's own
    stream, ordered behind the producing kernels) into buffer s % depth, `depth` = how long a prediction stays needed."""

    def __init__(self, rank, world, shape, depth, device, group=None):
        self.rank, self.world, self.group, self.depth = rank, world, group, depth
        self.send = torch.zeros((depth, 2) + tuple(shape), device=device, dtype=torch.float32)
        self.recv = torch.zeros((depth, world, 2) + tuple(shape), device=device, dtype=torch.float32)
        self.work = {}
        self.emulated = False

    def start(self, s, preds):
        import torch.distributed as dist
        d = s % self.depth
        for k, p in enumerate(preds):
            self.send[d, k].copy_(p)
        if dist.get_world_size(self.group) != self.world:
            # one rank of a larger world emulated on a single GPU (Pose2VideoPipeline.shard_emulate): the collective runs on the
            # real group with 1 / world of the bytes; the other ranks' entries are stand-ins
            self.emulated = True
            self.work[s] = dist.all_gather_into_tensor(self.recv[d, 0].view(-1), self.send[d].view(-1), group=self.group, async_op=True)
        elif self.send.is_cuda and dist.get_backend(self.group) == "gloo":  # no device collectives in gloo: host staging
            host = self.send[d].reshape(-1).cpu()
            out = torch.empty((self.world * host.numel(),), dtype=host.dtype)
            dist.all_gather_into_tensor(out, host, group=self.group)
            self.recv[d].copy_(out.view(self.recv[d].shape))
        else:
            self.work[s] = dist.all_gather_into_tensor(self.recv[d].view(-1), self.send[d].view(-1), group=self.group, async_op=True)

    def wait(self, s):
        w = self.work.pop(s, None)
        if w is not None:
            w.wait()
        if self.emulated:
            d = s % self.depth
            self.recv[d, 1:] = self.recv[d, :1]

    def unit(self, s, rank, k):
        return self.recv[s % self.depth, rank, k]

    def pair(self, s, rank):
        """both units of a whole-window item as one [2 Fw, h, w, C] prediction (uncond rows first)."""
        r = self.recv[s % self.depth, rank]
        return r.view((r.shape[0] * r.shape[1],) + tuple(r.shape[2:]))


def _all_gather(send, world, group=None):
    """all_gather of equal-shape tensors.  RCCL (backend "nccl") takes device tensors as they are; the gloo backend
    (CPU tests, and the 2-ranks-on-one-GPU test) has no device all_gather, so device tensors are staged through the host."""
    import torch.distributed as dist
    if dist.get_world_size(group) != world:
        # emulation of ONE rank of a larger world on a single GPU (Pose2VideoPipeline.shard_emulate): the collective runs on
        # the real (smaller) group — the backend's launch path is exercised — and the other ranks' chunks are stand-ins
        recv = [torch.empty_like(send) for _ in range(dist.get_world_size(group))]
        dist.all_gather(recv, send.contiguous(), group=group)
        return [recv[0]] * world
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

    Every rank owns as many slots as the busiest rank has units (assign_units).  `put(slot, pred)` copies the prediction into the rank's send
    slot and immediately starts the all-gather of THAT slot index over all ranks (async_op: RCCL runs it on its own
    stream, ordered after the producing kernels by an event), so the gather of slot k overlaps the compute of the
    rank's unit k+1.  `finish()` starts the gathers of slots this rank does not fill (it sends zeros there), waits
    for all of them and returns {unit: tensor}.  Buffers are allocated once per (shape, world) and reused every step."""

    def __init__(self, units, rank, world, shape, device, group=None):
        self.units, self.rank, self.world, self.group = units, rank, world, group
        self.assign = assign_units(units, world)
        self.nslots = 1 + max((slot for _, _, slot in self.assign), default=-1)  # the busiest rank's unit count
        self.send = torch.zeros((self.nslots,) + tuple(shape), device=device, dtype=torch.float32)
        self.recv = torch.zeros((self.nslots, world) + tuple(shape), device=device, dtype=torch.float32)
        self.my_slots = sum(1 for _, r, _ in self.assign if r == rank)
        self._host = None
        self.begin()

    def begin(self):
        self.works, self.started = [], 0

    def _gather_slot(self, k):
        import torch.distributed as dist
        if dist.get_world_size(self.group) != self.world:
            # one rank of a larger world emulated on a single GPU: the slot is gathered on the real group (same call, same
            # stream hand-over, 1 / world of the bytes); every other rank's entry is a copy of it (stand-in data)
            self.works.append(dist.all_gather_into_tensor(self.recv[k, 0].view(-1), self.send[k].view(-1), group=self.group,
                                                          async_op=True))
            self.emulated = True
            return
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
        if getattr(self, "emulated", False):
            self.recv[:, 1:] = self.recv[:, :1]
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

    def __init__(self, unet, b, F, h, w, c0, dtype, xdtype=None):
        dev = unet.device
        self.unet, self.b, self.F = unet, b, F
        self.x = torch.zeros((b * F, h, w, 8), device=dev, dtype=xdtype or dtype)   # (fp32 tokens when the input convolution takes split operands)
        self.pose = torch.zeros((b * F, h, w, c0), device=dev, dtype=torch.float32)
        # the step's rows of UNetBase.clip_tables (time-embedding projections, collapsed cross-attentions): the replay consumes
        # exactly what the eager run consumes, so the two are the same arithmetic bit for bit
        self.temb = self.attn2 = None
        self.graph = None
        self.out = None

    def _forward(self):
        return self.unet.run_tokens(self.x, None, None, self.b, self.F, self.pose, temb=self.temb, attn2=self.attn2)

    def capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture: weight packing, lazy library load
            self._forward()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        saved, ops.COUNTER = ops.COUNTER, {"flops": 0, "launches": 0}
        with torch.cuda.graph(self.graph):
            self.out = self._forward()
        self.work, ops.COUNTER = ops.COUNTER, saved  # algorithmic FLOPs / launches replayed by every graph launch

    def fingerprint(self):
        """Everything a captured graph bakes in by ADDRESS or by VALUE besides the static input buffers: the bank K/V
        buffers of the spatial blocks, the packed-weight generation (HipModule.invalidate drops packed buffers on
        .to() / load_state_dict) and the split-K setting.  A replay with a different fingerprint would read freed or
        stale memory, so the graph is re-captured instead."""
        from .modules import pack_epoch
        banks = tuple(0 if b.bank_kv is None else b.bank_kv.data_ptr() for b in self.unet.spatial_blocks())
        return (banks, pack_epoch(), ops.split_k_enabled())

    def __call__(self, x, temb, attn2, pose):
        """temb fp32 [b, sum(Cout)], attn2 fp32 [b, sum(C)]: the step's rows of clip_tables()."""
        if self.temb is None or self.temb.shape != temb.shape or self.attn2.shape != attn2.shape:
            self.temb, self.attn2, self.graph = torch.empty_like(temb), torch.empty_like(attn2), None
        self.x.copy_(x)
        self.pose.copy_(pose)
        self.temb.copy_(temb)
        self.attn2.copy_(attn2)
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
        self.dedup_frames = True  # encode each run of bit-identical consecutive input frames once (see _encode_frames)
        self.dist_group = None
        self.shard_windows = False  # True: deal (window, CFG half) units over the torch.distributed ranks
        self.shard_emulate = None   # (rank, world): run THAT rank's share of a `world`-GPU sharded clip on this one GPU (needs a
        #                             process group of size 1; other ranks' results are stand-ins: a timing mode, bench.py --emulate-rank)
        self.shard_force = False    # True: take the sharded code path (unit plan, per-slot async all_gather on the backend's
        #                             stream, item streams, sharded per-frame stages) even in a group of ONE rank — how the RCCL
        #                             branch is exercised on a single-GPU box (tests/test_models_gpu.py, bench.py --force-shard)
        self.shard_plan = "cross_step"  # how the sharded clip's forwards are scheduled: "cross_step" (plan_cross_step: slots of
        #                             (window, step) items, no per-step barrier) | "step_sync" (plan_items: the round 2-5 per-step plan)
        self.batch_invariant = False  # True: bit-identical to the sharded run of the same clip (split-K off)
        self.use_graphs = False     # True: replay the denoising forward as a captured hipGraph (single-GPU path)
        self.window_streams = 2     # > 1: the independent windows of one step run on this many HIP streams (single-GPU path)
        self._side_streams = {}
        self.stage_times = None     # dict -> accumulates per-stage milliseconds (HIP events) of run_tensors
        self._graphs = {}

    def _streams(self, dev):
        """The side streams of `dev` for the CURRENT window_streams setting (a later change of the attribute takes effect;
        each workspace slot these streams use pins one split-K scratch buffer in ops._WS, see ops.release_workspaces)."""
        key = (torch.device(dev), int(self.window_streams))
        st = self._side_streams.get(key)
        if st is None:
            st = self._side_streams[key] = [torch.cuda.Stream(dev) for _ in range(self.window_streams)]
        return st

    def prepack(self):
        """Pack every model's weights on the CURRENT stream (see HipModule.prepack): must precede any stream fork."""
        for m in (self.vae, self.pose_guider, self.reference_unet, self.denoising_unet):
            if hasattr(m, "prepack"):
                m.prepack(m.compute_dtype)

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
        -> fp32 latent tokens [n,h,w,4] * 0.18215 (pipeline :427-443).

        Runs of bit-identical consecutive frames are encoded ONCE (self.dedup_frames): run_animate.py's background is F
        copies of one frame (tools/util.py:339-345), which the reference encodes F times (:436-443).  One compare kernel
        and one F-integer read-back decide; the encoder is deterministic per image, so the clip's latents are unchanged."""
        dt = self.vae.compute_dtype
        from .image import ImageTokens
        tokens_in = isinstance(images, ImageTokens)
        split = getattr(self.vae, "encode_precision", "half") == "split"
        if tokens_in and images.dtype != dt and not (split and images.dtype == torch.float32):
            raise ValueError(f"pre-tokenised images are {images.dtype}, the VAE computes in {dt}"
                             + (" (fp32 tokens are the split policy's input)" if images.dtype == torch.float32 else ""))
        n = images.shape[0]
        plain = images.as_subclass(torch.Tensor) if tokens_in else images
        index = None
        if self.dedup_frames and n > 1:
            diff = ops.frames_differ(plain.contiguous())
            if diff is not None:
                starts = [1 if (d or i == 0) else 0 for i, d in enumerate(diff.tolist())]  # 1 = first frame of a run
                if sum(starts) < n:
                    run_of, k = [], -1
                    for s_ in starts:
                        k += s_
                        run_of.append(k)
                    index = torch.tensor(run_of, device=plain.device)
                    plain = plain[torch.tensor([i for i, s_ in enumerate(starts) if s_], device=plain.device)]
        hw = plain.shape[1:3] if tokens_in else plain.shape[-2:]
        nb = min(self.vae_batch, self.vae.max_images(*hw, split=split) if split else self.vae.max_images(*hw))
        outs = []
        for i in range(0, plain.shape[0], nb):
            chunk = plain[i:i + nb]
            if split:
                from .vae import nchw_to_tokens32
                # (pre-tokenised HALF input carries no low part: the image was rounded when it was tokenised; __call__ asks
                # image.vae_preprocess for fp32 tokens under this policy)
                x32 = chunk.float().contiguous() if tokens_in else nchw_to_tokens32(chunk)
                outs.append(self.vae.encode_tokens_split(x32))
            else:
                tok = chunk.contiguous() if tokens_in else ops.ncfhw_to_tokens(chunk.float().contiguous()[:, :, None], dt, cpad=8)
                outs.append(self.vae.encode_tokens(tok))
        lat = torch.cat(outs) * VAE_SCALE
        return lat if index is None else lat[index]

    def _decode_frames(self, latents):
        """latents fp32 [1,4,F,h,w] -> video fp32 [1,3,F,H,W] in [0,1] (decode_latents, pipeline :113-126)."""
        dt = self.vae.compute_dtype
        _, C, F, h, w = latents.shape
        split = getattr(self.vae, "decode_precision", "half") == "split"
        if split:
            from .vae import nchw_to_tokens32
            z = nchw_to_tokens32((latents[0] * (1.0 / VAE_SCALE)).permute(1, 0, 2, 3))  # fp32 [F,h,w,8]
        else:
            z = ops.ncfhw_to_tokens((latents * (1.0 / VAE_SCALE)).contiguous(), dt, cpad=8)  # [F,h,w,8]
        frames = []
        nb = min(self.vae_batch, self.vae.max_images(8 * h, 8 * w, split=split))
        for i in range(0, F, nb):
            y = (self.vae.decode_tokens_split if split else self.vae.decode_tokens)(z[i:i + nb].contiguous())
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
            if self.shard_emulate is not None:
                world = self.shard_emulate[1]
        with ops.split_k(not (world > 1 or self.batch_invariant or (self.shard_windows and self.shard_force))):
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
            if self.shard_emulate is not None:
                if world != 1:
                    raise ValueError("shard_emulate runs on a process group of size 1")
                rank, world = self.shard_emulate
        sharded = world > 1 or (self.shard_windows and self.shard_force and torch.distributed.is_available()
                                and torch.distributed.is_initialized())
        sched.set_timesteps(num_inference_steps)
        latents = latents.to(device=dev, dtype=torch.float32).contiguous().clone()
        _, C, F, h, w = latents.shape
        # lazily packed weights are packed HERE, on the main stream, before any side stream is forked: a side stream
        # that hit a cache entry published by another stream would read buffers it is not ordered behind
        self.prepack()
        steps_t = sched.timesteps.tolist()

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
            from .image import ImageTokens
            if isinstance(frames, ImageTokens):
                tok = frames.as_subclass(torch.Tensor)  # already half tokens (image.vae_preprocess)
                if tok.dtype != self.pose_guider.compute_dtype:
                    raise TypeError(f"pose image tokens are {tok.dtype}, the pose guider computes in {self.pose_guider.compute_dtype}")
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
        if not sharded and self.window_streams > 1:
            side = self._streams(dev)[0]
            side.wait_stream(main)
        # (fp32 tokens for the UNet input when its first convolution takes split operands: the split policy, or ops.EDGE_SPLIT)
        in32 = getattr(unet, "precision", "half") == "split" or bool(ops.EDGE_SPLIT & 4)
        bdt = torch.float32 if in32 else dt   # the background latents' token type
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()), ops.workspace_slot(1 if side is not None else 0):
            ref_tok = torch.zeros((1, h, w, 8), device=dev, dtype=torch.float32 if (getattr(self.reference_unet, "precision", "half") == "split"
                                                                                   or (ops.EDGE_SPLIT & 4)) else self.reference_unet.compute_dtype)
            ref_tok[..., :C] = ref_lat.to(ref_tok.dtype)
            rctx = Ctx(self.reference_unet.compute_dtype, 1, 1)
            rctx.stop_after = writer.last_block()
            try:
                self.reference_unet.run_tokens(ref_tok, 0, ehs_c, 1, 1, None, rctx)
            except EarlyExit:
                pass

        if sharded:  # one long clip over the ranks: the per-frame stages are sharded too
            bk_tok = sharded_frames(self._encode_frames, bk_images.to(dev), rank, world, self.dist_group).to(bdt)
            pose_tok = sharded_frames(pose_fn, pose_images.to(dev), rank, world, self.dist_group)
        else:
            bk_tok = self._encode_frames(bk_images.to(dev)).to(bdt)            # [F,h,w,4]
            pose_tok = pose_fn(pose_images.to(dev))                            # fp32 [F,h,w,C0]

        if side is not None:
            main.wait_stream(side)
            for blk in self.reference_unet.spatial_blocks():  # the banked tensors were allocated on the side stream
                for bt in blk.bank:
                    bt.record_stream(main)
        reader.update(writer)
        mark("vae_encode+pose_guider+reference_unet")

        # (the UNets' split precision policy, mimo_amd.precise, takes fp32 tokens: nothing of its input is rounded)
        xdt = torch.float32 if in32 else dt
        if getattr(unet, "precision", "half") == "split" and self.use_graphs:
            raise NotImplementedError("hipGraph replay covers the default precision policy only")
        windows = get_context_scheduler(context_schedule)(0, num_inference_steps, F, context_frames, context_stride,
                                                          context_overlap)
        win_idx = [torch.tensor(c, dtype=torch.int32, device=dev) for c in windows]
        win_pose = [pose_tok[c.long()] for c in win_idx]
        rep = (2 if cfg else 1) if not sharded else 1
        # Per-window UNet input [rep * Fw, h, w, 8] = (latent | background) channels, allocated once per clip: the
        # background half is constant and written here for every CFG copy, the latent half is rewritten every step by
        # the layout kernel (one launch per CFG copy) — no torch.cat / repeat inside the step loop.
        win_x = []
        for c in win_idx:
            xw = torch.empty((rep * c.numel(), h, w, 2 * C), device=dev, dtype=xdt)
            xw[..., C:] = bk_tok[c.long()].repeat(rep, 1, 1, 1)
            win_x.append(xw)
        if rep > 1:
            win_pose = [p_.repeat(rep, 1, 1, 1) for p_ in win_pose]

        def fill_latents(wi):
            idx, Fw = win_idx[wi], win_idx[wi].numel()
            for r_ in range(rep):
                fill_window_tokens(latents, idx, win_x[wi][r_ * Fw:(r_ + 1) * Fw], C, dt)
            return win_x[wi]

        # time-embedding projections of ALL steps and the collapsed cross-attentions of the clip: four GEMMs per clip
        # instead of four M = 2 launches per forward (they depend on (t, clip embedding) only)
        temb_tab, attn2_tab = unet.clip_tables(steps_t, ehs, 2 if cfg else 1)
        units, my_units = plan_units(len(windows), cfg, rank, world)
        my_items = plan_items(len(windows), cfg, world)[rank] if (sharded and self.shard_plan != "cross_step") else []
        item_x, item_pose = [], []
        for item in my_items:  # per-item UNet input / pose buffers of the sharded mode (same layout as win_x)
            c = win_idx[item[0][0]]
            xw = torch.empty((len(item) * c.numel(), h, w, 2 * C), device=dev, dtype=xdt)
            xw[..., C:] = bk_tok[c.long()].repeat(len(item), 1, 1, 1)
            item_x.append(xw)
            item_pose.append(pose_tok[c.long()].repeat(len(item), 1, 1, 1))
        counter_x = torch.zeros((F,), device=dev, dtype=torch.float32)  # the cond half's (unused) frame counter
        exch = None
        if sharded and self.shard_plan != "cross_step":  # every window has the same frame count; the UNet's output head is padded to 4 channels
            cpad = (unet.out_channels + 3) // 4 * 4
            exch = UnitExchange(units, rank, world, (len(windows[0]), h, w, cpad), dev, self.dist_group)
        gather_marks = []
        acc = torch.empty((2 if cfg else 1, C, F, h, w), device=dev, dtype=torch.float32)
        counter = torch.empty((F,), device=dev, dtype=torch.float32)

        cross = sharded and self.shard_plan == "cross_step"
        if cross:
            gather_marks = self._denoise_cross_step(latents, windows, win_idx, bk_tok, pose_tok, temb_tab, attn2_tab, ehs, steps_t,
                                                    cfg, guidance_scale, rank, world, callback, trajectory)
        for step, t in enumerate(() if cross else steps_t):
            tk = dict(temb=temb_tab[step], attn2=attn2_tab)
            acc.zero_()
            counter.zero_()
            preds = {}
            if not sharded and len(win_idx) > 1 and self.window_streams > 1 and not self.use_graphs:
                # The windows of one step are independent forwards: on two streams their kernels fill each other's idle
                # CUs (tail rounds of small levels, HBM-bound linears beside MFMA-bound convolutions): -4.7 % per step at
                # two windows (profiles/r2_two_stream_forward.txt).  Accumulation stays in canonical window order.
                main = torch.cuda.current_stream(dev)
                streams = self._streams(dev)
                for s_ in streams:
                    s_.wait_stream(main)
                wpred = []
                for wi, idx in enumerate(win_idx):
                    slot = wi % len(streams)
                    with torch.cuda.stream(streams[slot]), ops.workspace_slot(1 + slot):
                        wpred.append(unet.run_tokens(fill_latents(wi), t, ehs, rep, idx.numel(), win_pose[wi], **tk))
                for s_ in streams:
                    main.wait_stream(s_)
                for pred, idx in zip(wpred, win_idx):
                    pred.record_stream(main)
                    ops.window_accumulate(pred, idx, acc, counter)
            elif not sharded:
                for wi, idx in enumerate(win_idx):
                    x = fill_latents(wi)                                                         # [rep*Fw,h,w,8]
                    if self.use_graphs:
                        key = (rep, idx.numel(), h, w, dt, xdt)
                        if key not in self._graphs:
                            self._graphs[key] = GraphedDenoiser(unet, rep, idx.numel(), h, w, pose_tok.shape[-1], dt, xdt)
                        pred = self._graphs[key](x, tk["temb"], tk["attn2"], win_pose[wi])
                    else:
                        pred = unet.run_tokens(x, t, ehs, rep, idx.numel(), win_pose[wi], **tk)
                    ops.window_accumulate(pred, idx, acc, counter)
            else:
                # a rank's items (whole windows as b = 2 forwards, single halves as b = 1 forwards) are independent:
                # with two or more they run on alternating HIP streams (profiles/r2_two_stream_forward.txt); each
                # unit's prediction is handed to the exchange, in slot order, as soon as its stream is done
                conc = len(my_items) > 1 and self.window_streams > 1
                main = torch.cuda.current_stream(dev)
                streams = self._streams(dev) if conc else [main]
                if conc:
                    for s_ in streams:
                        s_.wait_stream(main)
                pending = []

                def hand_over(entry):
                    preds, s0 = entry
                    if conc:
                        main.wait_stream(s0)
                    for p0 in preds:
                        if conc:
                            p0.record_stream(main)
                        exch.put(p0)  # the gather of this unit's slot starts now and runs under the following items

                for k, item in enumerate(my_items):
                    wi = item[0][0]
                    idx, Fw = win_idx[wi], win_idx[wi].numel()
                    slot = k % len(streams)
                    with torch.cuda.stream(streams[slot]), ops.workspace_slot(1 + slot if conc else 0):
                        x = item_x[k]
                        for r_ in range(len(item)):
                            fill_window_tokens(latents, idx, x[r_ * Fw:(r_ + 1) * Fw], C, dt)
                        if len(item) == 2:  # the whole window, batched exactly as on one GPU
                            pred = unet.run_tokens(x, t, ehs, 2, Fw, item_pose[k], **tk)
                            preds = [pred[:Fw], pred[Fw:]]
                        else:
                            half = item[0][1]
                            e = ehs[half:half + 1] if cfg else ehs
                            hk = dict(temb=tk["temb"][half:half + 1], attn2=attn2_tab[half:half + 1])  # this half's rows
                            preds = [self._run_unit(unet, x, t, e, Fw, item_pose[k], cond=(half == 1 or not cfg), **hk)]
                    pending.append((preds, streams[slot]))
                    # the oldest pending item is handed over once every stream has a younger item queued behind it
                    if len(pending) == len(streams):
                        hand_over(pending.pop(0))
                for entry in pending:
                    hand_over(entry)
                if self.stage_times is not None:  # exposed part of the exchange: what the main stream waits for here
                    g0 = torch.cuda.Event(enable_timing=True)
                    g0.record()
                allp = exch.finish()
                if self.stage_times is not None:
                    g1 = torch.cuda.Event(enable_timing=True)
                    g1.record()
                    gather_marks.append((g0, g1))
                for wi, idx in enumerate(win_idx):  # canonical window order, uncond then cond: the single-GPU sums
                    for hf in ((0, 1) if cfg else (0,)):
                        ops.window_accumulate(allp[(wi, hf)], idx, acc[hf:hf + 1], counter if hf == 0 else counter_x)
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
        if sharded:
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
            if gather_marks:
                self.stage_times["exposed_gather_ms"] = self.stage_times.get("exposed_gather_ms", 0.0) + \
                    sum(g0.elapsed_time(g1) for g0, g1 in gather_marks)
        return (video, latents) if return_latents else video

    def _denoise_cross_step(self, latents, windows, win_idx, bk_tok, pose_tok, temb_tab, attn2_tab, ehs, steps_t, cfg,
                            guidance_scale, rank, world, callback, trajectory):
        """The denoising loop of the sharded long clip under the cross-step schedule (plan_cross_step): this rank runs its
        item of every slot, hands the prediction(s) to the slot's all_gather, and — like every other rank — advances the frames
        whose covering windows have all delivered.  Per element the arithmetic is the single-GPU loop's (canonical window
        order, mimo_cfg_ddim_step), so the clip reproduces the single-GPU bits."""
        dev = self.device
        unet, sched = self.denoising_unet, self.scheduler
        dt = unet.compute_dtype
        _, C, F, h, w = latents.shape
        T = len(steps_t)
        slots = best_cross_step_plan(windows, T, world, cfg)
        segs = frame_segments(windows, F)
        depth = cross_step_lifetime(slots, windows, F) + 1   # + 1: slot s + 1 is being written while slot s is still read
        Fw = len(windows[0])
        cpad = (unet.out_channels + 3) // 4 * 4
        exch = SlotExchange(rank, world, (Fw, h, w, cpad), depth, dev, self.dist_group)
        nb = 2 if cfg else 1
        acc = torch.empty((nb, C, F, h, w), device=dev, dtype=torch.float32)
        counter = torch.empty((F,), device=dev, dtype=torch.float32)
        counter_x = torch.zeros((F,), device=dev, dtype=torch.float32)  # the cond half's (unused) frame counter
        frame_step = [0] * F              # host mirror: the step each frame's latent is at
        where = {}                        # (w, t) -> [(slot, rank, halves)]: where the gathered prediction(s) sit
        in_x, in_pose = {}, {}            # per (window, rows) input buffers of this rank (background half written once)
        idx_cache = {}
        snaps = {} if (trajectory is not None or callback is not None) else None
        reported = 0
        gather_marks = []

        def dev_idx(key, values):
            t_ = idx_cache.get(key)
            if t_ is None:
                t_ = idx_cache[key] = torch.tensor(values, dtype=torch.int32, device=dev)
            return t_

        def item_input(wi, rows):
            key = (wi, rows)
            if key not in in_x:
                c = win_idx[wi].long()
                xw = torch.empty((rows * c.numel(), h, w, 2 * C), device=dev, dtype=bk_tok.dtype if bk_tok.dtype == torch.float32 else dt)
                xw[..., C:] = bk_tok[c].repeat(rows, 1, 1, 1)
                in_x[key], in_pose[key] = xw, pose_tok[c].repeat(rows, 1, 1, 1)
            return in_x[key], in_pose[key]

        def advance(s):
            """after slot s's gather: advance every frame segment that has become complete, lowest step first."""
            nonlocal reported
            for t in sorted({frame_step[fr[0]] for _, fr in segs}):
                if t >= T:
                    continue
                ready = [(cv, fr) for cv, fr in segs if frame_step[fr[0]] == t and all((v, t) in where for v in cv)]
                if not ready:
                    continue
                group = sorted(f for _, fr in ready for f in fr)
                gset = set(group)
                acc.zero_()
                counter.zero_()
                for wi in sorted({v for cv, _ in ready for v in cv}):       # canonical window order
                    mask = dev_idx(("m", wi, tuple(group)), [f if f in gset else -1 for f in windows[wi]])
                    for (s0, r0, halves) in where[(wi, t)]:
                        if len(halves) == 2:
                            ops.window_accumulate(exch.pair(s0, r0), mask, acc, counter)
                        elif not cfg:
                            ops.window_accumulate(exch.unit(s0, r0, 0), mask, acc, counter)
                        else:
                            hf = halves[0]
                            ops.window_accumulate(exch.unit(s0, r0, 0), mask, acc[hf:hf + 1], counter if hf == 0 else counter_x)
                gi = dev_idx(("g", tuple(group)), group)
                ops.cfg_ddim_step(acc, counter, latents, cfg, guidance_scale, *sched.coefficients(steps_t[t]), frames=gi)
                for f in group:
                    frame_step[f] = t + 1
                if snaps is not None:
                    if t not in snaps:
                        snaps[t] = torch.empty_like(latents)
                    snaps[t][:, :, gi.long()] = latents[:, :, gi.long()]
            while snaps is not None and reported < T and min(frame_step) > reported:
                snap = snaps.pop(reported)
                if trajectory is not None:
                    trajectory.append(snap)
                if callback is not None:
                    callback(reported, steps_t[reported], snap)
                reported += 1

        for s, slot in enumerate(slots):
            mine = [it for it in slot if it[0] == rank]
            preds = []
            for _, wi, t, halves in mine:
                idx, n_f = win_idx[wi], win_idx[wi].numel()
                assert all(frame_step[f] == t for f in windows[wi]), "cross-step plan ran ahead of its dependencies"
                x, pose = item_input(wi, len(halves))
                for r_ in range(len(halves)):
                    fill_window_tokens(latents, idx, x[r_ * n_f:(r_ + 1) * n_f], C, dt)
                if len(halves) == 2:
                    pred = unet.run_tokens(x, steps_t[t], ehs, 2, n_f, pose, temb=temb_tab[t], attn2=attn2_tab)
                    preds = [pred[:n_f], pred[n_f:]]
                else:
                    hf = halves[0]
                    e = ehs[hf:hf + 1] if cfg else ehs
                    preds = [self._run_unit(unet, x, steps_t[t], e, n_f, pose, cond=(hf == 1 or not cfg),
                                            temb=temb_tab[t][hf:hf + 1], attn2=attn2_tab[hf:hf + 1])]
            exch.start(s, preds)
            if self.stage_times is not None:
                g0 = torch.cuda.Event(enable_timing=True)
                g0.record()
            exch.wait(s)
            if self.stage_times is not None:
                g1 = torch.cuda.Event(enable_timing=True)
                g1.record()
                gather_marks.append((g0, g1))
            for r0, wi, t, halves in slot:
                where.setdefault((wi, t), []).append((s, r0, halves))
            advance(s)
            for key in [k for k, v in where.items() if all(s - s0 >= depth - 2 for s0, _, _ in v)]:
                # (ring entries two slots from being overwritten must have been consumed: every frame of the window is past t)
                assert all(frame_step[f] > key[1] for f in windows[key[0]]), "gather ring too shallow"
                del where[key]
        assert min(frame_step) == T and max(frame_step) == T
        return gather_marks


    def _run_unit(self, unet, x, t, ehs1, Fw, pose, cond, **tables):
        """One (window, CFG half) unit as a b = 1 forward.  The uncond half must not read the bank."""
        blocks = unet.spatial_blocks()
        saved = None
        if not cond:
            saved = [b.bank_kv for b in blocks]
            for b in blocks:
                b.bank_kv = None
        try:
            return unet.run_tokens(x, t, ehs1, 1, Fw, pose, **tables)
        finally:
            if saved is not None:
                for b, kv in zip(blocks, saved):
                    b.bank_kv = kv

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, ref_image, pose_images, vid_bk_images, width, height, video_length, num_inference_steps,
                 guidance_scale, num_images_per_prompt=1, eta=0.0, generator=None, output_type="tensor",
                 return_dict=True, callback=None, callback_steps=1, context_schedule="uniform", context_frames=24,
                 context_stride=1, context_overlap=4, context_batch_size=1, interpolation_factor=1, output_device=False,
                 **kwargs):
        """output_device=True (not in the reference): `.videos` stays an fp32 DEVICE tensor — what mimo_amd.run_edit hands to
        the device-side compositing instead of the reference's per-frame `.cpu().numpy()` round trips."""
        if eta != 0.0:
            raise NotImplementedError("eta = 0 (DDIM, the reference's setting) only")
        if context_batch_size != 1:
            # the reference itself cannot run this: with two windows per batch `encoder_hidden_states[:b]` has 2 rows for a
            # batch of 4 and `noise_pred[:, :, c] + pred` (:540) adds a 4-row prediction to a 2-row accumulator
            raise NotImplementedError("context_batch_size > 1 fails in the reference (shape mismatch at "
                                      "pipeline_pose2vid_long_edit_bkfill_roiclip.py:524,540); windows run one per forward")
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
        vdt = torch.float32 if getattr(self.vae, "encode_precision", "half") == "split" else self.vae.compute_dtype
        ref_t = IM.vae_preprocess([ref_image], height, width, True, vdt, dev)
        bk_t = IM.vae_preprocess(list(vid_bk_images), height, width, True, vdt, dev)
        pose_t = IM.vae_preprocess(list(pose_images), height, width, False, self.pose_guider.compute_dtype, dev)
        cb = None
        if callback is not None:
            cb = lambda i, t, lat: callback(i, t, lat) if i % callback_steps == 0 else None
        if interpolation_factor >= 2:  # :566-567: decode the frame-interpolated latents
            lat = self.run_tensors(ref_t, bk_t, pose_t, clip_embeds, latents.float(), num_inference_steps, guidance_scale,
                                   context_schedule, context_frames, context_stride, context_overlap, callback=cb, decode=False)
            video = self._decode_frames(interpolate_latents(lat, interpolation_factor))
        else:
            video = self.run_tensors(ref_t, bk_t, pose_t, clip_embeds, latents.float(), num_inference_steps, guidance_scale,
                                     context_schedule, context_frames, context_stride, context_overlap, callback=cb)
        if output_device:
            return Pose2VideoPipelineOutput(videos=video.float()) if return_dict else video.float()
        images = video.cpu().float()
        if output_type != "tensor":
            images = images.numpy()
        return Pose2VideoPipelineOutput(videos=images) if return_dict else images
