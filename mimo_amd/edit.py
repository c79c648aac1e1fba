"""Device-side post-processing of the video-editing entry point: the compositing loop of run_edit.py:253-304.

`composite_clips` is the drop-in for that loop: same inputs (the pipeline's `.videos[0]` tensor — which stays on the
device —, the clip contexts / bounding boxes / paddings computed by `crop_human_clip_auto_context` and `pad_img`, the
original background / video / occluder frames, the per-frame edge masks), same `res_images` result (uint8 [H, W, 3] per
frame), bit for bit.  Per generated frame it issues three launches: the two passes of the PIL-exact bicubic resize
(reading the fp32 video tensor in place, quantising as `(image * 255).astype(np.uint8)`) and ONE fused compositing
kernel (un-pad, paste, alpha blend, occluder, overlap cross-fade, truncation).  The reference instead copies every
frame to the host and runs PIL + NumPy + cv2 per frame.

`get_mask` (tools/util.py:397-447) is mirrored; the cv2 INTER_AREA resize of the selected mask to the clip size
(run_edit.py:284) is template preparation and stays with the caller (`masks[video_idx]`, float32 [h, w])."""
import numpy as np
import torch

from . import image as IM

MASK_MODE = {'up_down_left_right': 0, 'left_right_up': 1, 'left_right_down': 2, 'up_down_left': 3, 'up_down_right': 4,
             'left_right': 5, 'up_down': 6, 'left_up': 7, 'right_up': 8, 'left_down': 9, 'right_down': 10,
             'left': 11, 'right': 12, 'up': 13, 'down': 14, 'inner': 15}


def mask_mode(bbox, size):
    """Which of the 16 edge masks applies to a clip bounding box (tools/util.py:397-447 `get_mask`): sides of the box
    that touch the frame border keep a hard edge.  Returns the key of MASK_MODE."""
    w, h = size
    w_min, w_max, h_min, h_max = bbox
    L, R, U, D = w_min <= 0, w_max >= w, h_min <= 0, h_max >= h
    if L and R and U and D:
        return 'up_down_left_right'
    if L and R and U:
        return 'left_right_up'
    if L and R and D:
        return 'left_right_down'
    if L and U and D:
        return 'up_down_left'
    if R and U and D:
        return 'up_down_right'
    if L and R:
        return 'left_right'
    if U and D:
        return 'up_down'
    if L and U:
        return 'left_up'
    if R and U:
        return 'right_up'
    if L and D:
        return 'left_down'
    if R and D:
        return 'right_down'
    if L:
        return 'left'
    if R:
        return 'right'
    if U:
        return 'up'
    if D:
        return 'down'
    return 'inner'


def get_mask(mask_list, bbox, img):
    return mask_list[MASK_MODE[mask_mode(bbox, img.size)]]


def _frames_to_device(frames, device):
    """list of PIL / uint8 arrays / one uint8 tensor [L, H, W, 3] -> uint8 device tensor [L, H, W, 3]"""
    if frames is None:
        return None
    if torch.is_tensor(frames):
        return frames.to(device)
    return torch.from_numpy(np.stack([np.asarray(f.convert("RGB")) if hasattr(f, "convert") else np.asarray(f)
                                      for f in frames])).to(device)


def composite_clips(video, context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk_images_ori, vid_images_ori,
                    occ_mask_images, masks, overlay=4, L=None):
    """video: fp32 device tensor [3, Ftot, H, W] in [0, 1] (`pipe.run_tensors(...)[0]`, not copied to the host).
    Returns uint8 device tensor [L, Hf, Wf, 3]; frames no clip covers stay zero (the reference leaves None)."""
    dev = video.device
    assert video.dim() == 4 and video.shape[0] == 3 and video.dtype == torch.float32 and video.is_contiguous()
    _, Ftot, H, W = video.shape
    bk = _frames_to_device(bk_images_ori, dev)
    vid = _frames_to_device(vid_images_ori, dev)
    occ = _frames_to_device(occ_mask_images, dev)
    L = bk.shape[0] if L is None else L
    Hf, Wf = bk.shape[1], bk.shape[2]
    out = torch.zeros((L, Hf, Wf, 3), device=dev, dtype=torch.uint8)
    done = [False] * L
    video_idx = 0
    for k, context in enumerate(context_list):
        start_i = context[0]
        w_min, w_max, h_min, h_max = (int(v) for v in bbox_clip_list[k])
        for i in context:
            pad_h, pad_w = (int(v) for v in clip_pad_list[video_idx])
            # frame video_idx of the [3, Ftot, H, W] tensor, read in place: strides (image, row, column, channel)
            crop = IM.resize_u8(video[:, video_idx], (pad_h, pad_w), "bicubic", src_f32=True, src_hw=(H, W),
                                strides=(0, W, 1, Ftot * H * W), n=1)[0]
            m = masks[video_idx]
            m = m if torch.is_tensor(m) else torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32))
            m = m.to(device=dev, dtype=torch.float32).contiguous()
            factor = (i - start_i + 1) / (overlay + 1)
            IM.composite_frame(crop, clip_padv_list[video_idx], (w_min, h_min), m, bk[i], out[i],
                               occ=None if occ is None else occ[i], vid=None if occ is None else vid[i],
                               prev=out[i] if done[i] else None, factor=factor)
            done[i] = True
            video_idx += 1
    return out
