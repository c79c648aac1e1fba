"""Device-side image pre/post-processing around the denoising path.

Host code only decodes PIL images to raw uint8 bytes and builds small integer tables; every pixel operation runs in
libmimo_hip.so (mimo_amd/csrc/image.hip):

  resize_u8          PIL.Image.resize (BICUBIC / LANCZOS) bit for bit: Pillow's 8-bit two-pass resampler with its
                     fixed-point coefficient tables (built here exactly as Resample.c precompute_coeffs +
                     normalize_coeffs_8bpc do) applied by mimo_resample_pass_u8.
  vae_preprocess     diffusers VaeImageProcessor.preprocess as configured by the reference pipeline
                     (pipeline_pose2vid_long_edit_bkfill_roiclip.py:73-80,424-457): RGB, LANCZOS resize to (width, height)
                     rounded down to a multiple of 8, /255, optional 2x-1 -> half tokens [n, H, W, 8].
  clip_preprocess    `ref_image.resize((224, 224))` + CLIPImageProcessor (:379-384): bicubic resize, shortest-edge resize,
                     centre crop, rescale 1/255, normalise -> fp32 [1, 3, 224, 224].
  composite_frame    one frame of run_edit.py:253-304 (un-pad, paste, alpha blend, occluder, overlap cross-fade).
"""
import ctypes
import math
from functools import lru_cache

import numpy as np
import torch

from . import lib as L
from . import ops

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # OPENAI_CLIP_MEAN / STD (transformers CLIPImageProcessor defaults)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


@lru_cache(maxsize=256)
def pil_coeffs(in_size, out_size, filt):
    """Pillow's precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc: (bounds int32 [out, 2], kk int32 [out, ksize])."""
    f, sup = _FILTERS[filt]
    in0, in1 = np.float32(0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = sup * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


_DEV_TABLES = {}


def _tables(in_size, out_size, filt, device):
    key = (in_size, out_size, filt, str(device))
    t = _DEV_TABLES.get(key)
    if t is None:
        b, k = pil_coeffs(in_size, out_size, filt)
        t = _DEV_TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
    return t


def resize_u8(src, out_hw, filt="bicubic", *, src_f32=False, src_hw=None, strides=None, n=None, C=3):
    """PIL.Image.resize((W', H'), resample=filt) of a batch of images, bit for bit.

    src: uint8 [n, H, W, C] (contiguous), or with src_f32=True an fp32 tensor addressed by element `strides`
    (image, row, column, channel) and `src_hw`, quantised on the fly as (v * 255).astype(uint8).  Returns uint8 [n, H', W', C]."""
    ops._chk(src, "src")
    Ho, Wo = out_hw
    if strides is None:
        assert src.dim() == 4 and src.is_contiguous() and src.dtype == torch.uint8
        n, H, W, C = src.shape
        strides = (H * W * C, W * C, C, 1)
    else:
        H, W = src_hw
    dev = src.device
    cur, cur_f32, cur_strides, cur_hw = src, src_f32, strides, (H, W)
    st = ops._stream()
    if W != Wo:  # Pillow: horizontal pass first
        b, k, ks = _tables(W, Wo, filt, dev)
        tmp = torch.empty((n, H, Wo, C), device=dev, dtype=torch.uint8)
        L.call("mimo_resample_pass_u8", cur.data_ptr(), int(cur_f32), *cur_strides, tmp.data_ptr(), n, H, Wo, C,
               b.data_ptr(), k.data_ptr(), ks, 1, st)
        cur, cur_f32, cur_strides, cur_hw = tmp, False, (H * Wo * C, Wo * C, C, 1), (H, Wo)
    if H != Ho:
        b, k, ks = _tables(H, Ho, filt, dev)
        out = torch.empty((n, Ho, Wo, C), device=dev, dtype=torch.uint8)
        L.call("mimo_resample_pass_u8", cur.data_ptr(), int(cur_f32), *cur_strides, out.data_ptr(), n, Ho, Wo, C,
               b.data_ptr(), k.data_ptr(), ks, 0, st)
        cur, cur_f32 = out, False
    if cur_f32 or cur is src:  # same size: Image.resize returns a copy; an fp32 source still needs its quantisation
        b, k, ks = _identity_tables(W, dev)
        out = torch.empty((n, H, W, C), device=dev, dtype=torch.uint8)
        L.call("mimo_resample_pass_u8", cur.data_ptr(), int(cur_f32), *cur_strides, out.data_ptr(), n, H, W, C,
               b.data_ptr(), k.data_ptr(), ks, 1, st)
        cur = out
    return cur


def _identity_tables(size, device):
    key = (size, "identity", str(device))
    t = _DEV_TABLES.get(key)
    if t is None:
        b = np.stack([np.arange(size, dtype=np.int32), np.ones(size, np.int32)], 1)
        k = np.full((size, 1), 1 << PRECISION_BITS, np.int32)
        t = _DEV_TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), 1)
    return t


def pil_to_u8(images, device):
    """PIL images of one size -> uint8 [n, H, W, 3] on `device`: decode + ONE host-to-device copy of the raw bytes."""
    arr = np.stack([np.asarray(im.convert("RGB")) for im in images])
    return torch.from_numpy(arr).to(device)


class ImageTokens(torch.Tensor):
    """Marker type of pre-tokenised images (half — or fp32 — tokens [n, h, w, 8] from vae_preprocess): the pipeline recognises them by
    TYPE, not by shape and dtype — a half NCHW batch of width 8 is a legal image input and must not be mistaken for tokens."""

    @staticmethod
    def wrap(t):
        return t.as_subclass(ImageTokens)


def vae_preprocess(images, height, width, normalize, dtype, device, scale_factor=8):
    """list of PIL images -> half tokens [n, h, w, 8] (channels 3..7 zero), as VaeImageProcessor.preprocess does it.
    dtype torch.float32: the same values unrounded (the input of the VAE encoder's split-operand policy)."""
    if len(images) == 0:
        raise ValueError("vae_preprocess needs at least one image (got an empty list)")
    w, h = width - width % scale_factor, height - height % scale_factor
    groups, out = {}, [None] * len(images)
    for i, im in enumerate(images):  # images of one size share a launch
        groups.setdefault(im.size, []).append(i)
    for size, idx in groups.items():
        u8 = pil_to_u8([images[i] for i in idx], device)
        if (size[0], size[1]) != (w, h):
            u8 = resize_u8(u8, (h, w), "lanczos")
        tok = torch.empty((len(idx), h, w, 8), device=device, dtype=dtype)
        L.call("mimo_u8_to_tokens", L.F32 if dtype == torch.float32 else ops.dt_code(dtype), u8.data_ptr(), len(idx) * h * w, 3, 8, int(bool(normalize)),
               tok.data_ptr(), ops._stream())
        for j, i in enumerate(idx):
            out[i] = tok[j:j + 1]
    return ImageTokens.wrap(torch.cat(out) if len(groups) > 1 else tok)


def clip_preprocess(image, device, size=224):
    """`ref_image.resize((224, 224))` (PIL default BICUBIC) -> CLIPImageProcessor: resize shortest edge to 224 (bicubic; a
    no-op on a 224 x 224 input), centre crop 224 (no-op), rescale 1/255, normalise with the CLIP mean / std.
    Returns fp32 [1, 3, 224, 224] on `device`."""
    u8 = pil_to_u8([image], device)
    if tuple(u8.shape[1:3]) != (size, size):
        u8 = resize_u8(u8, (size, size), "bicubic")
    mean = torch.tensor(CLIP_MEAN, device=device, dtype=torch.float32)
    std = torch.tensor(CLIP_STD, device=device, dtype=torch.float32)
    out = torch.empty((1, 3, size, size), device=device, dtype=torch.float32)
    L.call("mimo_u8_to_planar_f32", u8.data_ptr(), 1, size * size, 3, 1.0 / 255.0, mean.data_ptr(), std.data_ptr(),
           out.data_ptr(), ops._stream())
    return out


def composite_frame(crop, padding_v, paste_xy, mask, bk, out, *, occ=None, vid=None, prev=None, factor=0.0):
    """One frame of run_edit.py:253-304.  crop: uint8 [pad_h, pad_w, 3] (the generated frame resized to the padded clip
    size); padding_v = (top, bottom, left, right); paste_xy = (w_min, h_min); mask: fp32 [mh, mw] (already resized);
    bk / occ / vid / prev / out: uint8 [H, W, 3] device tensors."""
    ops._chk(crop, "crop")
    p = L.CompositeParams()
    p.crop, p.mask, p.bk, p.out = crop.data_ptr(), mask.data_ptr(), bk.data_ptr(), out.data_ptr()
    p.occ, p.vid, p.prev = ops._ptr(occ), ops._ptr(vid), ops._ptr(prev)
    p.factor = float(factor)
    p.pad_h, p.pad_w = crop.shape[0], crop.shape[1]
    p.top, p.bottom, p.left, p.right = (int(v) for v in padding_v)
    p.w_min, p.h_min = int(paste_xy[0]), int(paste_xy[1])
    p.mh, p.mw = mask.shape
    p.H, p.W = out.shape[0], out.shape[1]
    for t in (crop, bk, out) + tuple(x for x in (occ, vid, prev) if x is not None):
        assert t.dtype == torch.uint8 and t.is_contiguous()
    assert mask.dtype == torch.float32 and mask.is_contiguous()
    L.call("mimo_composite_frame", ctypes.byref(p), ops._stream())
    return out
