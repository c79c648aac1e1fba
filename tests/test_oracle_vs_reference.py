"""Tier-1 pin: the self-contained oracle (oracle/models.py, oracle/pipeline.py) equals the reference's OWN
code (/root/reference/src run behind oracle/diffusers_standin.py) on seeded weights, CPU fp32.
Runs only where /root/reference is mounted (this container); the GPU box uses the oracle + tests/golden/."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.reference

MM_KW = dict(use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
             use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=True,
             motion_module_decoder_only=False, motion_module_type="Vanilla",
             motion_module_kwargs=dict(num_attention_heads=4, num_transformer_block=1,
                                       attention_block_types=["Temporal_Self", "Temporal_Self"],
                                       temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                       temporal_attention_dim_div=1))  # configs/inference/inference_v2.yaml:1-22 (4 heads: half-width test model)


@pytest.fixture(scope="module")
def ref():
    from oracle.diffusers_standin import install
    install()
    import src.models.unet_3d_edit_bkfill as u3
    import src.models.unet_2d_condition as u2
    import src.models.pose_guider as pg
    import src.models.mutual_self_attention as msa
    import src.pipelines.context as ctx
    return dict(u3=u3, u2=u2, pg=pg, msa=msa, ctx=ctx)


def test_context_scheduler_equals_reference(ref):
    from oracle.pipeline import uniform
    for F in [1, 8, 24, 25, 26, 48, 64, 150, 192]:
        assert uniform(0, 20, F, 24, 1, 4) == list(ref["ctx"].uniform(0, 20, F, 24, 1, 4))


def test_state_dict_layout_full_size(ref):
    """Key names + shapes of the oracle trees == the reference trees at the real SD1.5 size (meta device)."""
    from oracle import models as OM
    kw = dict(sample_size=64, cross_attention_dim=768, attention_head_dim=8)
    mm = dict(MM_KW, motion_module_kwargs=dict(MM_KW["motion_module_kwargs"], num_attention_heads=8))
    with torch.device("meta"):
        r3 = ref["u3"].UNet3DConditionModel(in_channels=8, **kw, **mm)
        r2 = ref["u2"].UNet2DConditionModel(in_channels=4, **kw)
        rp = ref["pg"].PoseGuider(320, 3, (16, 32, 96, 256))
        o3 = OM.UNet3DConditionModel()
        o2 = OM.UNet2DConditionModel()
        op = OM.PoseGuider()
    for r, o, n in ((r3, o3, 1274), (r2, o2, 682), (rp, op, 16)):
        rs = {k: tuple(v.shape) for k, v in r.state_dict().items()}
        os_ = {k: tuple(v.shape) for k, v in o.state_dict().items()}
        assert rs == os_ and len(rs) == n
    assert sum(p.numel() for p in o3.parameters()) == 1312741764
    assert sum(p.numel() for p in o2.parameters()) == 859508800


def test_pose_guider_equals_reference(ref):
    from oracle import models as OM, synth
    o = synth.build(OM.PoseGuider, 7)
    r = ref["pg"].PoseGuider(320, 3, (16, 32, 96, 256)).eval()
    r.load_state_dict(o.state_dict(), strict=True)
    x = torch.rand(1, 3, 3, 32, 32)
    with torch.no_grad():
        assert torch.allclose(o(x), r(x), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("hw,F", [(16, 4), (13, 3)])  # 13: odd latent size -> forward_upsample_size path (784 default)
def test_unets_and_bank_equal_reference(ref, hw, F):
    from oracle import models as OM, synth
    kw = synth.small_unet_kwargs()
    o3 = synth.build(OM.UNet3DConditionModel, 11, motion_heads=4, **kw)
    o2 = synth.build(OM.UNet2DConditionModel, 12, **kw)
    r3 = ref["u3"].UNet3DConditionModel(sample_size=hw, in_channels=8, **kw, **MM_KW).eval()
    r2 = ref["u2"].UNet2DConditionModel(sample_size=hw, in_channels=4, **kw).eval()
    assert r3.load_state_dict(o3.state_dict(), strict=True)
    assert r2.load_state_dict(o2.state_dict(), strict=True)
    g = torch.Generator().manual_seed(5)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    ref_lat = torch.randn(1, 4, hw, hw, generator=g)
    x = torch.randn(2, 8, F, hw, hw, generator=g)
    pose = torch.randn(2, 160, F, hw, hw, generator=g)
    t = torch.tensor(749)
    with torch.no_grad():
        w_r = ref["msa"].ReferenceAttentionControl(r2, do_classifier_free_guidance=True, mode="write", batch_size=1, fusion_blocks="full")
        rd_r = ref["msa"].ReferenceAttentionControl(r3, do_classifier_free_guidance=True, mode="read", batch_size=1, fusion_blocks="full")
        r2(ref_lat.repeat(2, 1, 1, 1), torch.zeros_like(t), encoder_hidden_states=ehs, return_dict=False)
        rd_r.update(w_r)
        out_r = r3(x, t, encoder_hidden_states=ehs, pose_cond_fea=pose, return_dict=False)[0]
        w_o = OM.ReferenceAttentionControl(o2, "write")
        rd_o = OM.ReferenceAttentionControl(o3, "read")
        o2(ref_lat.repeat(2, 1, 1, 1), torch.zeros_like(t), ehs)
        rd_o.update(w_o)
        out_o = o3(x, t, ehs, pose_cond_fea=pose)
    # banks: same tensors, same pairing
    r_blocks = [m for m in ref["msa"].torch_dfs(r3) if type(m).__name__ == "TemporalBasicTransformerBlock"]
    r_blocks = sorted(r_blocks, key=lambda b: -b.norm1.normalized_shape[0])
    for rb, ob in zip(r_blocks, o3.spatial_blocks()):
        assert rb.bank[0].dtype == torch.float16 and torch.equal(rb.bank[0], ob.bank[0])
    assert out_r.shape == out_o.shape == (2, 4, F, hw, hw)
    assert torch.allclose(out_o, out_r, atol=2e-5, rtol=1e-4), float((out_o - out_r).abs().max())


def test_pipeline_equals_reference(ref):
    """Reference Pose2VideoPipeline.__call__ vs oracle.pipeline.run_clip: F = 26 -> two wrapped 24-frame windows."""
    import numpy as np
    from PIL import Image
    from oracle import models as OM, synth, primitives as P
    from oracle.pipeline import run_clip
    from src.pipelines.pipeline_pose2vid_long_edit_bkfill_roiclip import Pose2VideoPipeline
    kw = synth.small_unet_kwargs()
    H = W = 64
    F = 26
    o3 = synth.build(OM.UNet3DConditionModel, 21, motion_heads=4, **kw)
    o2 = synth.build(OM.UNet2DConditionModel, 22, **kw)
    opg = synth.build(OM.PoseGuider, 23, conditioning_embedding_channels=160)
    vae = synth.build(P.AutoencoderKL, 24, block_out_channels=(32, 32, 64, 64), norm_num_groups=8)
    r3 = ref["u3"].UNet3DConditionModel(sample_size=8, in_channels=8, **kw, **MM_KW).eval()
    r2 = ref["u2"].UNet2DConditionModel(sample_size=8, in_channels=4, **kw).eval()
    rpg = ref["pg"].PoseGuider(160, 3, (16, 32, 96, 256)).eval()
    r3.load_state_dict(o3.state_dict())
    r2.load_state_dict(o2.state_dict())
    rpg.load_state_dict(opg.state_dict())

    class FakeClip(torch.nn.Module):  # image encoder stub: deterministic embedding, fp32
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.emb = torch.randn(1, 768, generator=torch.Generator().manual_seed(3))

        @property
        def dtype(self):
            return torch.float32

        def forward(self, x):
            return type("O", (), {"image_embeds": self.emb})()

    clip = FakeClip()
    sched = P.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS)
    pipe = Pose2VideoPipeline(vae=vae, image_encoder=clip, reference_unet=r2, denoising_unet=r3, pose_guider=rpg, scheduler=sched)
    rs = np.random.RandomState(0)
    ref_img = Image.fromarray(rs.randint(0, 256, (H, W, 3), dtype=np.uint8))
    poses = [Image.fromarray(np.random.RandomState(100 + i).randint(0, 256, (H, W, 3), dtype=np.uint8)) for i in range(F)]
    bks = [Image.fromarray(np.random.RandomState(200 + i).randint(0, 256, (H, W, 3), dtype=np.uint8)) for i in range(F)]
    gen = torch.manual_seed(42)
    out_r = pipe(ref_img, poses, bks, W, H, F, 2, 3.5, generator=gen).videos
    # oracle on tensors
    to_t = lambda im: torch.from_numpy(np.array(im).astype(np.float32) / 255.0).permute(2, 0, 1)
    lat = torch.randn((1, 4, F, H // 8, W // 8), generator=torch.Generator().manual_seed(42))
    out_o, _ = run_clip(vae, o2, o3, opg, P.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip.emb,
                        (2 * to_t(ref_img) - 1)[None], torch.stack([2 * to_t(b) - 1 for b in bks]),
                        torch.stack([to_t(p) for p in poses]), lat, 2, 3.5)
    assert out_r.shape == out_o.shape == (1, 3, F, H, W)
    assert torch.allclose(out_o, out_r, atol=1e-4), float((out_o - out_r).abs().max())
