"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by mimo_amd/, never on the product path).

CPU fp32 restatement, in plain PyTorch, of the third-party arithmetic the reference's hot
path delegates to `diffusers==0.24.0` (pinned in /root/reference/install.sh:12; NOT vendored
under /root/reference, NOT installed in this image).  The published v0.24.0 semantics are
restated here and anchored on the reference's own call sites (cited per class).  The
reference ships no tests / golden vectors for this path, so parity is
pinned by (i) running the reference's own `src/` against these primitives (oracle/diffusers
stand-in, Tier-1) and (ii) golden tensors frozen from that run under tests/golden/.
"""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# diffusers.models.attention_processor.{Attention, AttnProcessor, AttnProcessor2_0}
# call sites: src/models/attention.py:109-141,321-345; src/models/motion_module.py:282-292,379-385
# ----------------------------------------------------------------------------------------
class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        bsz = query.shape[0]
        hd = query.shape[-1] // attn.heads
        query = query.view(bsz, -1, attn.heads, hd).transpose(1, 2)
        key = key.view(bsz, -1, attn.heads, hd).transpose(1, 2)
        value = value.view(bsz, -1, attn.heads, hd).transpose(1, 2)
        # softmax(q k^T / sqrt(d)) v, no mask, no dropout
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(bsz, -1, attn.heads * hd).to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


AttnProcessor = AttnProcessor2_0  # same arithmetic (explicit softmax vs fused SDPA)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, added_kv_proj_dim=None,
                 norm_num_groups=None, spatial_norm_dim=None, out_bias=True, scale_qk=True,
                 only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None, **_unused):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True) if norm_num_groups else None
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor or AttnProcessor2_0()

    def set_processor(self, processor, **_):
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, *a, **k):
        pass

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        kw.pop("video_length", None)
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


# ----------------------------------------------------------------------------------------
# diffusers.models.attention.{GEGLU, FeedForward}
# call sites: src/models/attention.py:152-157,359,429; src/models/motion_module.py:235,258
# ----------------------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, hidden_states, scale=1.0):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)  # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        assert activation_fn == "geglu", "only GEGLU is reached by the SD1.5 / motion-module configs"
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])

    def forward(self, hidden_states, scale=1.0):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


# ----------------------------------------------------------------------------------------
# diffusers.models.embeddings.{Timesteps, TimestepEmbedding}
# call sites: src/models/unet_3d_edit_bkfill.py:94-97,462-468; src/models/unet_2d_condition.py:320-335
# ----------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, scale=1.0,
                           max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


def get_activation(name):
    return {"silu": nn.SiLU(), "swish": nn.SiLU(), "gelu": nn.GELU(), "relu": nn.ReLU(), "mish": nn.Mish()}[name]


# ----------------------------------------------------------------------------------------
# diffusers.models.lora.LoRACompatible{Conv,Linear}: plain conv / linear that ignore `scale`
# ----------------------------------------------------------------------------------------
class LoRACompatibleConv(nn.Conv2d):
    def forward(self, x, scale=1.0):
        return super().forward(x)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, x, scale=1.0):
        return super().forward(x)


# ----------------------------------------------------------------------------------------
# diffusers.models.resnet.{ResnetBlock2D, Downsample2D, Upsample2D}
# call sites: src/models/unet_2d_blocks.py:391-445,547-599,704-731,820-865,988-1007 and the VAE
# ----------------------------------------------------------------------------------------
class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.name = channels, out_channels or channels, use_conv, name
        assert not use_conv_transpose
        if use_conv:
            self.conv = LoRACompatibleConv(channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None, scale=1.0):
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        if self.use_conv:
            hidden_states = self.conv(hidden_states)
        return hidden_states


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.padding = channels, out_channels or channels, use_conv, padding
        assert use_conv
        self.conv = LoRACompatibleConv(channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states, scale=1.0):
        if self.use_conv and self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None,
                 up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.output_scale_factor = in_channels, out_channels, output_scale_factor
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups_out, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = LoRACompatibleConv(out_channels, conv_2d_out_channels or out_channels, 3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.use_in_shortcut = in_channels != (conv_2d_out_channels or out_channels) if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = LoRACompatibleConv(in_channels, conv_2d_out_channels or out_channels, 1, stride=1,
                                                    padding=0, bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb, scale=1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        if self.time_emb_proj is not None and temb is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


# ----------------------------------------------------------------------------------------
# diffusers.models.autoencoder_kl.AutoencoderKL (sd-vae-ft-mse config)
# call sites: run_animate.py:70-72; src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:71,120,430,438
# ----------------------------------------------------------------------------------------
class _VaeMid(nn.Module):
    def __init__(self, ch, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(ch, heads=1, dim_head=ch, rescale_output_factor=1.0, eps=eps,
                                                   norm_num_groups=groups, residual_connection=True, bias=True,
                                                   upcast_softmax=True, _from_deprecated_attn_block=True)])
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=ch, out_channels=ch, temb_channels=None, eps=eps, groups=groups),
                                      ResnetBlock2D(in_channels=ch, out_channels=ch, temb_channels=None, eps=eps, groups=groups)])

    def forward(self, x):
        x = self.resnets[0](x, None)
        x = self.attentions[0](x)
        return self.resnets[1](x, None)


class _VaeDownBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_down, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout,
                                                    temb_channels=None, eps=eps, groups=groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, use_conv=True, out_channels=cout, padding=0)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _VaeUpBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=cin if i == 0 else cout, out_channels=cout,
                                                    temb_channels=None, eps=eps, groups=groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, use_conv=True, out_channels=cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x, None)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups, eps=1e-6):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            self.down_blocks.append(_VaeDownBlock(c, co, layers_per_block, i != len(block_out_channels) - 1, groups, eps))
            c = co
        self.mid_block = _VaeMid(c, groups, eps)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, 2 * out_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups, eps=1e-6):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = _VaeMid(rev[0], groups, eps)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_VaeUpBlock(c, co, layers_per_block + 1, i != len(rev) - 1, groups, eps))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)

    def mode(self):
        return self.mean


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getitem__(self, i):
        return list(self.__dict__.values())[i]


class _Cfg(dict):
    __getattr__ = dict.__getitem__


VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                  latent_channels=4, norm_num_groups=32, act_fn="silu", sample_size=256, scaling_factor=0.18215)


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, act_fn="silu", sample_size=256, scaling_factor=0.18215, **_):
        super().__init__()
        self.config = _Cfg(in_channels=in_channels, out_channels=out_channels, block_out_channels=list(block_out_channels),
                           layers_per_block=layers_per_block, latent_channels=latent_channels,
                           norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, list(block_out_channels), layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, list(block_out_channels), layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_slicing(self):
        pass

    def disable_slicing(self):
        pass

    def encode(self, x, return_dict=True):
        return _Out(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z, return_dict=True):
        return _Out(sample=self.decoder(self.post_quant_conv(z)))


# ----------------------------------------------------------------------------------------
# diffusers.schedulers.DDIMScheduler (kwargs: configs/inference/inference_v2.yaml:24-33)
# call sites: run_animate.py:96-97; pipeline :373,182,519-521,551-553
# ----------------------------------------------------------------------------------------
def rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    s = alphas_cumprod.sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    s = (s - sT) * s0 / (s0 - sT)
    alphas_bar = s ** 2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 timestep_spacing="leading", rescale_betas_zero_snr=False, **_):
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        self.prediction_type, self.timestep_spacing, self.clip_sample = prediction_type, timestep_spacing, clip_sample
        self.num_inference_steps = None
        import numpy as np
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        import numpy as np
        self.num_inference_steps = num_inference_steps
        T = self.num_train_timesteps
        if self.timestep_spacing == "trailing":
            timesteps = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        elif self.timestep_spacing == "leading":
            ratio = T // num_inference_steps
            timesteps = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        elif self.timestep_spacing == "linspace":
            timesteps = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(self.timestep_spacing)
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        assert eta == 0.0
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError(self.prediction_type)
        if self.clip_sample:
            x0 = x0.clamp(-1, 1)
        direction = (1 - a_prev) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        return _Out(prev_sample=prev, pred_original_sample=x0)


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: a CPU generator draws on CPU then moves."""
    gdev = generator.device.type if generator is not None and not isinstance(generator, list) else "cpu"
    if gdev == "cpu":
        return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)
