"""ORACLE — TEST INFRASTRUCTURE ONLY.  NumPy / PIL restatement of the post-processing loop of the reference's video
editing entry point, /root/reference/run_edit.py:253-304 (MIMO.run): for every clip and frame, quantise the generated
frame, PIL-resize it to the padded clip size, remove the padding, paste it on a white canvas at the clip's bounding box,
alpha-blend with the inpainted background through the edge mask, re-impose the occluder, cross-fade the `overlay` frames
two consecutive clips share, truncate to uint8.

run_edit.py cannot be imported here (cv2, imageio, tensorflow are absent), so this follows the cited lines statement
by statement with the same NumPy / PIL calls (the PIL resize is the real PIL: it pins the device resampler bit for
bit).  Not restated: `get_mask` + `cv2.resize(mask, ..., INTER_AREA)` (run_edit.py:283-284, tools/util.py:397-447) — the
resized float32 mask of every frame is an INPUT here and of the device op (`masks[video_idx]`), cv2 being unavailable.
"""
import numpy as np
from PIL import Image


def composite(video, context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk_images_ori, vid_images_ori,
              occ_mask_images, masks, overlay, L):
    """video: torch fp32 [3, Ftot, H, W] in [0, 1] (`pipe(...).videos[0]`); context_list: frame indices of every clip;
    bbox_clip_list[k] = (w_min, w_max, h_min, h_max); clip_pad_list / clip_padv_list / masks: per generated frame;
    bk / vid / occ: lists of PIL frames (occ may be None).  Returns the list res_images of uint8 [H, W, 3] arrays."""
    video_idx = 0
    res_images = [None for _ in range(L)]                                            # :255
    for k, context in enumerate(context_list):                                       # :256
        start_i = context[0]
        bbox = bbox_clip_list[k]
        for i in context:
            bk_image_pil_ori = bk_images_ori[i]
            vid_image_pil_ori = vid_images_ori[i]
            occ_mask = occ_mask_images[i] if occ_mask_images is not None else None   # :261-264
            canvas = Image.new("RGB", bk_image_pil_ori.size, "white")                # :266
            pad_h, pad_w = clip_pad_list[video_idx]
            padding_v = clip_padv_list[video_idx]
            image = video[:, video_idx, :, :].permute(1, 2, 0).cpu().numpy()         # :271
            res_image_pil = Image.fromarray((image * 255).astype(np.uint8))
            res_image_pil = res_image_pil.resize((pad_w, pad_h))                     # :273 (PIL default: BICUBIC)
            top, bottom, left, right = padding_v
            res_image_pil = res_image_pil.crop((left, top, pad_w - right, pad_h - bottom))
            w_min, w_max, h_min, h_max = bbox
            canvas.paste(res_image_pil, (w_min, h_min))                              # :279
            mask_full = np.zeros((bk_image_pil_ori.size[1], bk_image_pil_ori.size[0]), dtype=np.float32)
            mask = masks[video_idx]                                                  # :283-284 (input, see header)
            mask_full[h_min:h_min + mask.shape[0], w_min:w_min + mask.shape[1]] = mask
            res_image = np.array(canvas)
            bk_image = np.array(bk_image_pil_ori)
            res_image = res_image * mask_full[:, :, np.newaxis] + bk_image * (1 - mask_full[:, :, np.newaxis])   # :289
            if occ_mask is not None:
                vid_image = np.array(vid_image_pil_ori)
                occ = np.array(occ_mask)[:, :, 0].astype(np.uint8)
                occ = occ / 255.0
                res_image = res_image * (1 - occ[:, :, np.newaxis]) + vid_image * occ[:, :, np.newaxis]          # :295
            if res_images[i] is None:
                res_images[i] = res_image
            else:
                factor = (i - start_i + 1) / (overlay + 1)
                res_images[i] = res_images[i] * (1 - factor) + res_image * factor                               # :301
            res_images[i] = res_images[i].astype(np.uint8)
            video_idx = video_idx + 1
    return res_images
