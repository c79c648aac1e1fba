"""ORACLE — TEST INFRASTRUCTURE ONLY.  Seeded synthetic weights/inputs: no pretrained
checkpoints exist offline, so every parity check uses random-init weights of the real architecture, with
zero-initialised tensors re-drawn and norm affines perturbed so that no branch is numerically invisible."""
import torch
import torch.nn as nn

NOISE_SCHEDULER_KWARGS = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                              steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                              timestep_spacing="trailing")  # configs/inference/inference_v2.yaml:24-33


def randomize_(module: nn.Module, seed: int):
    """In place: N(0, 0.02) for all-zero weight tensors, gamma = 1 + 0.1 N, beta = 0.1 N for norms, biases 0.02 N."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in module.named_modules():
            if isinstance(m, (nn.GroupNorm, nn.LayerNorm)):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            elif isinstance(m, (nn.Conv2d, nn.Linear)):
                if float(m.weight.abs().max()) == 0.0:
                    m.weight.copy_(0.02 * torch.randn(m.weight.shape, generator=g))
                if m.bias is not None:
                    m.bias.copy_(0.02 * torch.randn(m.bias.shape, generator=g))
    return module


def build(cls, seed, **kw):
    torch.manual_seed(seed)
    m = cls(**kw)
    randomize_(m, seed + 1)
    return m.eval().requires_grad_(False)


def small_unet_kwargs():
    """A narrow UNet (same topology/heads/groups-per-channel structure) for second-scale CPU oracles."""
    # half width, 4 heads -> head dims 40/80/160/160 as in SD1.5; 32 groups (the motion modules hard-code 32)
    return dict(block_out_channels=(160, 320, 640, 640), norm_num_groups=32, attention_head_dim=4, cross_attention_dim=768)
