"""CPU-side checks (no GPU): C-ABI symbols, state-dict layout, host logic (windows, scheduler, packing,
work decomposition + world_size-2 exchange over gloo)."""
import ctypes
import math
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mimo_amd import build, lib
    path = build.build()
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, "include", "mimo_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t) (mimo_\w+)\(", header, flags=re.M))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    cdll = lib.load()
    for name in declared:
        assert hasattr(cdll, name)
    assert cdll.mimo_version() >= 1


def test_ops_fail_loudly_without_gpu():
    from mimo_amd import lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    a = torch.zeros(8, 8, dtype=torch.float16)
    with pytest.raises(lib.MimoHipError):
        ops.gemm(a, a)


def test_state_dict_layout_matches_oracle_full_size():
    from mimo_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    from mimo_amd.vae import AutoencoderKL, PoseGuider
    from oracle import models as OM, primitives as OP
    with torch.device("meta"):
        pairs = [(UNet3DConditionModel(), OM.UNet3DConditionModel(), 1274), (UNet2DConditionModel(), OM.UNet2DConditionModel(), 682),
                 (PoseGuider(), OM.PoseGuider(), 16), (AutoencoderKL(), OP.AutoencoderKL(), 248)]
    for p, o, n in pairs:
        ps = {k: tuple(v.shape) for k, v in p.state_dict().items()}
        os_ = {k: tuple(v.shape) for k, v in o.state_dict().items()}
        assert ps == os_ and len(ps) == n, (type(p).__name__, set(ps) ^ set(os_))


def test_vae_accepts_deprecated_attention_keys():
    from mimo_amd.vae import AutoencoderKL
    m = AutoencoderKL(block_out_channels=(32, 32), norm_num_groups=8)
    sd = m.state_dict()
    old = {}
    for k, v in sd.items():
        k2 = k.replace(".to_q.", ".query.").replace(".to_k.", ".key.").replace(".to_v.", ".value.").replace(".to_out.0.", ".proj_attn.")
        old[k2] = v.clone()
    m2 = AutoencoderKL(block_out_channels=(32, 32), norm_num_groups=8)
    m2.load_state_dict(old, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_context_windows_match_oracle():
    from mimo_amd.context import uniform
    from oracle.pipeline import uniform as ref
    for F in [1, 8, 24, 25, 26, 48, 64, 150, 192]:
        for step, stride in [(0, 1), (3, 3)]:
            assert uniform(step, 20, F, 24, stride, 4) == ref(step, 20, F, 24, stride, 4)
    w = uniform(0, 20, 192, 24, 1, 4)
    assert len(w) == 10 and w[0] == list(range(24)) and w[-1] == [(180 + i) % 192 for i in range(24)]


def test_scheduler_matches_oracle():
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    a, b = DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS)
    for S in (4, 20, 25):
        a.set_timesteps(S)
        b.set_timesteps(S)
        assert a.timesteps.tolist() == b.timesteps.tolist()
    assert a.timesteps.tolist()[:3] == [999, 959, 919]
    a.set_timesteps(20)
    b.set_timesteps(20)
    assert a.timesteps.tolist() == [999 - 50 * i for i in range(20)]
    assert float(a.alphas_cumprod[999]) == 0.0  # zero terminal SNR
    x, v = torch.randn(1, 4, 3, 5, 5), torch.randn(1, 4, 3, 5, 5)
    for t in (999, 499, 49):
        sa, s1, sap, s1p = a.coefficients(t)
        x0 = sa * x - s1 * v
        eps = sa * v + s1 * x
        assert torch.allclose(sap * x0 + s1p * eps, b.step(v, t, x).prev_sample, atol=1e-6)


def test_packing_layouts():
    from mimo_amd.packing import pack_conv, pack_geglu
    w = torch.randn(6, 5, 3, 3)
    p = pack_conv(w, torch.float32, cin_pad=8, cout_pad=8)
    assert p.shape == (8, 72) and torch.equal(p[2].reshape(3, 3, 8)[1, 2, :5], w[2, :, 1, 2]) and float(p[6:].abs().sum()) == 0
    wl, bl = torch.randn(64, 8), torch.randn(64)
    wp, bp = pack_geglu(wl, bl, torch.float32)
    assert torch.equal(wp[:16], wl[:16]) and torch.equal(wp[16:32], wl[32:48]) and torch.equal(wp[32:48], wl[16:32])
    assert torch.equal(bp[16:32], bl[32:48])


def test_plan_units_covers_everything_once():
    from mimo_amd.pipeline import plan_units
    for nw, cfg, world in [(1, True, 1), (1, True, 2), (10, True, 8), (3, False, 2), (5, True, 4)]:
        seen = []
        for r in range(world):
            units, mine = plan_units(nw, cfg, r, world)
            seen += mine
        assert sorted(seen) == sorted(units) and len(units) == nw * (2 if cfg else 1)


def _exchange_worker(rank, world, port, q):
    import torch.distributed as dist
    from mimo_amd.pipeline import exchange_predictions, plan_units
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    units, mine = plan_units(5, True, rank, world)
    fake = lambda u: torch.full((3, 2), float(10 * u[0] + u[1]))
    allp = exchange_predictions({u: fake(u) for u in mine}, units, rank, world)
    ok = all(torch.equal(allp[u], fake(u)) for u in units)
    # canonical-order window sum is rank independent
    total = sum(allp[u] for u in units)
    q.put((rank, ok, float(total.sum())))
    dist.destroy_process_group()


def test_exchange_predictions_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2]


def test_clip_image_encoder_state_dict_matches_transformers():
    """mimo_amd.clip mirrors transformers.CLIPVisionModelWithProjection key for key (ViT-L/14 shapes on the meta device)."""
    from transformers import CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as RefCLIP
    from mimo_amd.clip import CLIPVisionModelWithProjection
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                           image_size=224, patch_size=14, projection_dim=768)
    with torch.device("meta"):
        ref, prod = RefCLIP(cfg), CLIPVisionModelWithProjection(cfg)
    rs = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    ps = {k: tuple(v.shape) for k, v in prod.state_dict().items()}
    assert rs == ps
    assert sum(v.numel() for v in prod.parameters()) == sum(v.numel() for v in ref.parameters())


def _sharded_frames_worker(rank, world, port, q):
    import torch.distributed as dist
    from mimo_amd.pipeline import sharded_frames
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ok = True
    for F in (1, 5, 8):  # F < world, ragged, even
        x = torch.arange(F * 6, dtype=torch.float32).reshape(F, 2, 3)
        fn = lambda t: (t * 2 + 1).sum(dim=1)  # a per-frame op: [f, 2, 3] -> [f, 3]
        out = sharded_frames(fn, x, rank, world)
        ok &= bool(torch.equal(out, fn(x)))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_sharded_frames_world2_gloo():
    """Per-frame stages of the long-clip mode (VAE encode / decode, pose guider): contiguous frame chunks per rank +
    one all_gather reproduce the unsharded result, including F < world and ragged F."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500 + 7
    procs = [ctx.Process(target=_sharded_frames_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p_ in procs:
        p_.join(timeout=60)
    assert res == [(0, True), (1, True)]
