"""Shared builders for model-level parity tests: oracle (CPU fp32) and product (HIP) models with identical
seeded weights."""
import torch

from oracle import models as OM
from oracle import primitives as OP
from oracle import synth

MM4 = dict(num_attention_heads=4, temporal_position_encoding_max_len=32)


def small_kw():
    return synth.small_unet_kwargs()


def build_pair_unets(dtype, dev, seed=31):
    """(oracle3d, oracle2d, product3d, product2d) at half width (160/320/640/640, 4 heads -> d = 40/80/160)."""
    from mimo_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    kw = small_kw()
    o3 = synth.build(OM.UNet3DConditionModel, seed, motion_heads=4, **kw)
    o2 = synth.build(OM.UNet2DConditionModel, seed + 1, **kw)
    p3 = UNet3DConditionModel(motion_module_kwargs=MM4, **kw)
    p2 = UNet2DConditionModel(**kw)
    p3.load_state_dict(o3.state_dict(), strict=True)
    p2.load_state_dict(o2.state_dict(), strict=True)
    p3.to(dev)
    p2.to(dev)
    p3.compute_dtype = p2.compute_dtype = dtype
    return o3, o2, p3, p2


def build_pair_vae(dtype, dev, seed=41, boc=(32, 64, 64, 64), groups=8):
    from mimo_amd.vae import AutoencoderKL
    ov = synth.build(OP.AutoencoderKL, seed, block_out_channels=boc, norm_num_groups=groups)
    pv = AutoencoderKL(block_out_channels=boc, norm_num_groups=groups)
    pv.load_state_dict(ov.state_dict(), strict=True)
    pv.to(dev)
    pv.compute_dtype = dtype
    return ov, pv


def build_pair_pose(dtype, dev, seed=51, cout=160):
    from mimo_amd.vae import PoseGuider
    og = synth.build(OM.PoseGuider, seed, conditioning_embedding_channels=cout)
    pg = PoseGuider(conditioning_embedding_channels=cout)
    pg.load_state_dict(og.state_dict(), strict=True)
    pg.to(dev)
    pg.compute_dtype = dtype
    return og, pg
