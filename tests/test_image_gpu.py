"""Byte-exact image kernels through the C-ABI: the PIL-exact resampler against PIL itself, the
device-side VaeImageProcessor / CLIPImageProcessor paths against the host paths they replace, and the run_edit.py
compositing loop against its NumPy / PIL restatement (oracle/edit.py).  Integer / byte work: torch.equal, no tolerance."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _img(seed, h, w):
    return Image.fromarray(np.random.RandomState(seed).randint(0, 256, (h, w, 3), dtype=np.uint8))


@pytest.mark.parametrize("filt,pil", [("bicubic", Image.BICUBIC), ("lanczos", Image.LANCZOS)])
@pytest.mark.parametrize("hw,out_hw", [((64, 48), (80, 96)), ((100, 120), (53, 37)), ((64, 64), (224, 224)), ((90, 70), (64, 64)),
                                       ((512, 512), (476, 500)), ((37, 64), (37, 128)), ((784, 784), (512, 512))])
def test_resize_u8_equals_pil(dev, filt, pil, hw, out_hw):
    from mimo_amd import image as IM
    ims = [_img(s, *hw) for s in (1, 2, 3)]
    got = IM.resize_u8(IM.pil_to_u8(ims, dev), out_hw, filt).cpu().numpy()
    for g, im in zip(got, ims):
        ref = np.asarray(im.resize((out_hw[1], out_hw[0]), resample=pil))
        assert np.array_equal(g, ref)


def test_resize_from_fp32_video_tensor_in_place(dev):
    """The compositing path reads frame f of the pipeline's [3, F, H, W] fp32 video in place and quantises as
    (image * 255).astype(np.uint8) (run_edit.py:267-269)."""
    from mimo_amd import image as IM
    F, H, W = 3, 40, 56
    video = torch.rand(3, F, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    video[0, 1, 0, 0], video[1, 1, 0, 1] = 1.0, 0.0
    for f, (ph, pw) in enumerate([(64, 80), (40, 56), (33, 47)]):
        got = IM.resize_u8(video[:, f], (ph, pw), "bicubic", src_f32=True, src_hw=(H, W), strides=(0, W, 1, F * H * W), n=1)[0]
        image = video[:, f].permute(1, 2, 0).cpu().numpy()
        ref = np.asarray(Image.fromarray((image * 255).astype(np.uint8)).resize((pw, ph)))
        assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("normalize", [True, False])
def test_vae_preprocess_equals_host_path(dev, dtype, normalize):
    """Device VaeImageProcessor (decode -> bytes -> LANCZOS resize -> /255 -> 2x-1 -> half tokens) == the host path it
    replaces (PIL resize + numpy, then the layout kernel), bit for bit; mixed input sizes incl. the no-resize case."""
    from mimo_amd import image as IM, ops
    from oracle.diffusers_standin import VaeImageProcessor  # the reference's host path (PIL LANCZOS + numpy)
    ims = [_img(1, 96, 80), _img(2, 64, 64), _img(3, 96, 80), _img(4, 130, 70)]
    H = W = 64
    got = IM.vae_preprocess(ims, H, W, normalize, dtype, dev)
    proc = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True, do_normalize=normalize)
    host = torch.cat([proc.preprocess(im, height=H, width=W) for im in ims]).to(dev)
    ref = ops.ncfhw_to_tokens(host.contiguous()[:, :, None], dtype, cpad=8)
    assert got.shape == ref.shape == (4, H, W, 8) and torch.equal(got, ref)


def test_clip_preprocess_matches_transformers(dev):
    from transformers import CLIPImageProcessor
    from mimo_amd import image as IM
    im = _img(5, 300, 200)
    ref = CLIPImageProcessor().preprocess(im.resize((224, 224)), return_tensors="pt").pixel_values
    got = IM.clip_preprocess(im, dev).cpu()
    assert got.shape == ref.shape == (1, 3, 224, 224)
    assert float((got - ref.float()).abs().max()) < 2e-6  # same uint8 pixels; float normalisation order may differ by an ulp


@pytest.mark.parametrize("with_occ", [True, False])
def test_composite_clips_equals_run_edit_loop(dev, with_occ):
    """Two ROI clips with different boxes / paddings that share `overlay` = 4 frames (cross-fade), soft edge masks,
    optional occluder: uint8 result identical to the NumPy / PIL restatement of run_edit.py:253-304."""
    from mimo_amd import edit as E
    from oracle import edit as OE
    rs = np.random.RandomState(0)
    Hf, Wf, L, H, W, overlay = 96, 128, 12, 64, 64, 4
    bk = [_img(10 + i, Hf, Wf) for i in range(L)]
    vid = [_img(40 + i, Hf, Wf) for i in range(L)]
    occ = None
    if with_occ:
        occ = []
        for i in range(L):
            o = np.zeros((Hf, Wf, 3), np.uint8)
            o[20:50, 30 + i:70 + i] = rs.randint(0, 256, (30, 40, 1))
            occ.append(Image.fromarray(o))
    context_list = [list(range(0, 8)), list(range(4, 12))]          # frames 4..7 belong to both clips
    bbox_clip_list = [(10, 74, 8, 88), (40, 128, 0, 96)]            # (w_min, w_max, h_min, h_max)
    clip_pad_list, clip_padv_list, masks = [], [], []
    for k, ctx in enumerate(context_list):
        w_min, w_max, h_min, h_max = bbox_clip_list[k]
        cw, ch = w_max - w_min, h_max - h_min
        side = max(cw, ch)                                           # pad_img: pad the crop to a square
        top, left = (side - ch) // 2, (side - cw) // 2
        padv = (top, side - ch - top, left, side - cw - left)
        m = np.clip(rs.rand(ch, cw).astype(np.float32) * 1.2, 0, 1)  # stands for cv2.resize(get_mask(...), INTER_AREA)
        for _ in ctx:
            clip_pad_list.append([side, side])
            clip_padv_list.append(padv)
            masks.append(m)
    Ftot = sum(len(c) for c in context_list)
    video = torch.rand(3, Ftot, H, W, generator=torch.Generator().manual_seed(1))
    ref = OE.composite(video, context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk, vid, occ, masks, overlay, L)
    got = E.composite_clips(video.to(dev).contiguous(), context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk, vid,
                            occ, masks, overlay, L).cpu().numpy()
    for i in range(L):
        assert np.array_equal(got[i], ref[i]), (i, int(np.abs(got[i].astype(int) - ref[i].astype(int)).max()))
