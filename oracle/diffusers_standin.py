"""ORACLE — TEST INFRASTRUCTURE ONLY.  Tier-1 oracle support: a minimal in-memory stand-in for the
`diffusers` package (pinned 0.24.0 by /root/reference/install.sh:12, absent from this image) so the
reference's OWN `src/models/*.py` and `src/pipelines/*.py` import and run unmodified from
/root/reference on CPU fp32.  Only the names the reference imports are provided;
the arithmetic lives in oracle/primitives.py, everything else is plumbing or an import-only
placeholder for code paths the SD1.5 config never reaches.

    from oracle.diffusers_standin import install; install()   # then `import src.models...`
"""
import dataclasses
import inspect
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import primitives as P

REFERENCE_ROOT = "/root/reference"


class BaseOutput:
    """dataclass-style output: attribute access, integer indexing over non-None fields, to_tuple()."""

    def to_tuple(self):
        return tuple(getattr(self, f.name) for f in dataclasses.fields(self) if getattr(self, f.name) is not None)

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return self.to_tuple()[k]


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k not in ("self", "kwargs")}
        self._internal_dict = FrozenDict(cfg)
        init(self, *args, **kwargs)

    wrapper.__wrapped__ = init
    return wrapper


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kw)
        self._internal_dict = FrozenDict(d)

    @classmethod
    def load_config(cls, path, **_):
        path = str(path)
        if os.path.isdir(path):
            path = os.path.join(path, cls.config_name)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        init = getattr(cls.__init__, "__wrapped__", cls.__init__)
        names = set(inspect.signature(init).parameters) - {"self"}
        args = {k: v for k, v in dict(config).items() if k in names}
        args.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**args)


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        path = str(path)
        if subfolder:
            path = os.path.join(path, subfolder)
        model = cls.from_config(cls.load_config(path))
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        sd = {k: v for k, v in sd.items() if k in model.state_dict()}
        model.load_state_dict(sd, strict=False)
        model.eval()
        return model


class DiffusionPipeline:
    def register_modules(self, **kw):
        self._modules_names = list(kw)
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device=None, dtype=None):
        for k in self._modules_names:
            m = getattr(self, k)
            if isinstance(m, nn.Module):
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        for k in self._modules_names:
            m = getattr(self, k)
            if isinstance(m, nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        class _Bar:
            def __enter__(s):
                return s

            def __exit__(s, *a):
                return False

            def update(s, n=1):
                pass

        return _Bar()


class VaeImageProcessor:
    """RGB convert, PIL LANCZOS resize to (width, height) rounded down to x8, /255, NCHW, optional 2x-1."""

    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_binarize=False, do_convert_rgb=False, do_convert_grayscale=False):
        self.vae_scale_factor, self.do_normalize, self.do_convert_rgb = vae_scale_factor, do_normalize, do_convert_rgb

    def preprocess(self, image, height=None, width=None):
        from PIL import Image
        images = image if isinstance(image, list) else [image]
        out = []
        for im in images:
            if self.do_convert_rgb:
                im = im.convert("RGB")
            w, h = (width or im.width), (height or im.height)
            w, h = (x - x % self.vae_scale_factor for x in (w, h))
            im = im.resize((w, h), resample=Image.LANCZOS)
            out.append(np.array(im).astype(np.float32) / 255.0)
        t = torch.from_numpy(np.stack(out, 0)).permute(0, 3, 1, 2)
        return 2.0 * t - 1.0 if self.do_normalize else t


class _Logger:
    def __getattr__(self, k):
        return lambda *a, **kw: None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def _placeholder(name):
    def _init(self, *a, **k):
        raise NotImplementedError(f"diffusers.{name} is an import-only placeholder (not reached by the SD1.5 config)")

    return type(name, (nn.Module,), {"__init__": _init})


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(reference_root=REFERENCE_ROOT):
    """Register the stand-in as `diffusers` and put the reference checkout on sys.path."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "__mimo_standin__", False):
        return
    ph = {n: _placeholder(n) for n in [
        "AdaLayerNorm", "SinusoidalPositionalEmbedding", "CaptionProjection", "GaussianFourierProjection",
        "ImageHintTimeEmbedding", "ImageProjection", "ImageTimeEmbedding", "PositionNet", "TextImageProjection",
        "TextImageTimeEmbedding", "TextTimeEmbedding", "AdaLayerNormSingle", "DualTransformer2DModel",
        "AttnAddedKVProcessor", "PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler",
        "EulerAncestralDiscreteScheduler", "DPMSolverMultistepScheduler"]}
    utils = _mod("diffusers.utils", BaseOutput=BaseOutput, USE_PEFT_BACKEND=False,
                 deprecate=lambda *a, **k: None, is_torch_version=lambda op, v: True,
                 is_accelerate_available=lambda: False, logging=_Logging,
                 scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
                 SAFETENSORS_WEIGHTS_NAME="diffusion_pytorch_model.safetensors", WEIGHTS_NAME="diffusion_pytorch_model.bin")
    utils.import_utils = _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    utils.torch_utils = _mod("diffusers.utils.torch_utils", randn_tensor=P.randn_tensor,
                             apply_freeu=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))
    cfgu = _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    ap = _mod("diffusers.models.attention_processor", Attention=P.Attention, AttnProcessor=P.AttnProcessor,
              AttnProcessor2_0=P.AttnProcessor2_0, AttentionProcessor=object,
              AttnAddedKVProcessor=ph["AttnAddedKVProcessor"], ADDED_KV_ATTENTION_PROCESSORS=(),
              CROSS_ATTENTION_PROCESSORS=(P.AttnProcessor2_0,))
    att = _mod("diffusers.models.attention", FeedForward=P.FeedForward, AdaLayerNorm=ph["AdaLayerNorm"],
               Attention=P.Attention, GEGLU=P.GEGLU)
    emb = _mod("diffusers.models.embeddings", Timesteps=P.Timesteps, TimestepEmbedding=P.TimestepEmbedding,
               **{k: ph[k] for k in ["SinusoidalPositionalEmbedding", "CaptionProjection", "GaussianFourierProjection",
                                     "ImageHintTimeEmbedding", "ImageProjection", "ImageTimeEmbedding", "PositionNet",
                                     "TextImageProjection", "TextImageTimeEmbedding", "TextTimeEmbedding"]})
    res = _mod("diffusers.models.resnet", ResnetBlock2D=P.ResnetBlock2D, Downsample2D=P.Downsample2D, Upsample2D=P.Upsample2D)
    lora = _mod("diffusers.models.lora", LoRACompatibleConv=P.LoRACompatibleConv, LoRACompatibleLinear=P.LoRACompatibleLinear)
    act = _mod("diffusers.models.activations", get_activation=P.get_activation)
    norm = _mod("diffusers.models.normalization", AdaLayerNormSingle=ph["AdaLayerNormSingle"])
    dual = _mod("diffusers.models.dual_transformer_2d", DualTransformer2DModel=ph["DualTransformer2DModel"])
    mu = _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    models = _mod("diffusers.models", ModelMixin=ModelMixin, attention_processor=ap, attention=att, embeddings=emb,
                  resnet=res, lora=lora, activations=act, normalization=norm, dual_transformer_2d=dual,
                  modeling_utils=mu)
    loaders = _mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    sched = _mod("diffusers.schedulers", DDIMScheduler=P.DDIMScheduler,
                 **{k: ph[k] for k in ["PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler",
                                       "EulerAncestralDiscreteScheduler", "DPMSolverMultistepScheduler"]})
    imgp = _mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    top = _mod("diffusers", DiffusionPipeline=DiffusionPipeline, AutoencoderKL=P.AutoencoderKL,
               DDIMScheduler=P.DDIMScheduler, utils=utils, configuration_utils=cfgu, models=models, loaders=loaders,
               schedulers=sched, image_processor=imgp, __version__="0.24.0", __mimo_standin__=True)
    top.__path__ = []
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


def reference_available(reference_root=REFERENCE_ROOT):
    return os.path.isdir(os.path.join(reference_root, "src", "models"))
