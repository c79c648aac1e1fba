"""ROI-clip segmentation and padding of a video-editing template — the host-side data preparation either side of the
denoising path in run_edit.py.  Mirrors, with the reference's names and return values:

  extract_mask_sdc, clean_mask, crop_img_sdc, crop_img      tools/util.py:17-26,41-69
  pad_img                                                   tools/util.py:27-39
  init_bbox, bbox_div2, bbox_pad, compute_area_ratio        tools/util.py:111-160
  update_clip, crop_human_clip_auto_context                 tools/util.py:161-285
  crop_human, init_bk                                       tools/util.py:71-110,339-344 (the animate entry: one box for the clip)
  prepare_clips      the per-frame padding loop of run_edit.py:226-248 (pose / background lists handed to the pipeline)
  clip_masks         get_mask + cv2.resize(mask, crop size, INTER_AREA) of run_edit.py:283-284, per generated frame

This is integer / index logic on uint8 frames exactly as the reference runs it (NumPy on the host); the OpenCV calls
go through mimo_amd.cvops.  Not here: mp4 decode / encode (tools/util.py:462-479: codec work, no library in this image)
and the TensorFlow matting graph (tools/human_segmenter.py).  The outputs feed Pose2VideoPipeline.__call__ and
mimo_amd.edit.composite_clips (the device-side compositing), see INTEGRATION.md."""
import numpy as np
from PIL import Image

from . import cvops
from .edit import get_mask


def crop_img(img, mask):
    x, y, w, h = cvops.bounding_rect(mask)
    y_max, x_max = y + h, x + w
    y = max(0, y - int(h * 0.05))
    y_max = min(img.shape[0], y_max + int(h * 0.05))
    return img[y:y_max, x:x_max]


def pad_img(img, color=(255, 255, 255)):
    """pad to a square whose side is a multiple of 16; returns (image, [top, bottom, left, right])."""
    h, w = img.shape[:2]
    max_size = max(h, w)
    if max_size % 16 != 0:
        max_size = int(max_size / 16) * 16 + 16
    top = (max_size - h) // 2
    bottom = max_size - h - top
    left = (max_size - w) // 2
    right = max_size - w - left
    return cvops.copy_make_border(img, top, bottom, left, right, list(color)), [top, bottom, left, right]


def extract_mask_sdc(img):
    mask = np.zeros_like(img[:, :, 0])
    gray = cvops.rgb2gray(img)
    mask[gray[:, :] > 10] = 255
    return mask


def clean_mask(mask):
    return cvops.morphology_rect(cvops.morphology_rect(mask, "close", 5), "open", 2)


def crop_img_sdc(img, mask):
    x, y, w, h = cvops.bounding_rect(mask)
    y_max, x_max = y + h, x + w
    pad_h, pad_w = 0.1, 0.05
    y = max(0, y - int(h * pad_h))
    y_max = min(img.shape[0], y_max + int(h * pad_h))
    x = max(0, x - int(w * pad_w))
    x_max = min(img.shape[1], x_max + int(w * pad_w))
    return y, y_max, x, x_max


def crop_human(pose_images, vid_images, mask_images):
    """tools/util.py:71-110: ONE bounding box of the subject over all pose frames (per-frame crop_img_sdc boxes united, an odd
    side grown by one at its far edge — not re-clipped: a box at the image border stays odd after the slice, as in the
    reference), applied to the pose, video and background frames alike.  -> three lists of PIL."""
    y, y_max, x, x_max = 10000, 0, 10000, 0
    for pose_img in pose_images:
        frame = np.array(pose_img)
        y_, y_max_, x_, x_max_ = crop_img_sdc(frame, extract_mask_sdc(frame))
        y, y_max, x, x_max = min(y, y_), max(y_max, y_max_), min(x, x_), max(x_max, x_max_)
    if (y_max - y) % 2 == 1:
        y_max += 1
    if (x_max - x) % 2 == 1:
        x_max += 1
    cut = lambda images: [Image.fromarray(np.array(im)[y:y_max, x:x_max]) for im in images]
    return cut(pose_images), cut(vid_images), cut(mask_images)


def init_bk(n_frame, h, w):
    """tools/util.py:339-344: n white frames of h x w."""
    return [Image.fromarray(np.ones((h, w, 3), dtype=np.uint8) * 255) for _ in range(n_frame)]


def init_bbox():
    return [10000, 0, 10000, 0]


def bbox_div2(x, x_max, y, y_max):
    if (y_max - y) % 2 == 1:
        y_max += 1
    if (x_max - x) % 2 == 1:
        x_max += 1
    return x, x_max, y, y_max


def bbox_pad(x, x_max, y, y_max, img):
    w, h = x_max - x, y_max - y
    max_size = max(h, w)
    if max_size % 16 != 0:
        max_size = int(max_size / 16) * 16 + 16
    top = (max_size - h) // 2
    bottom = max_size - h - top
    left = (max_size - w) // 2
    right = max_size - w - left
    return max(0, x - left), min(img.shape[1], x_max + right), max(0, y - top), min(img.shape[0], y_max + bottom)


def compute_area_ratio(bbox_frame, bbox_clip):
    x1, x2, y1, y2 = bbox_frame
    a1, a2, b1, b2 = bbox_clip
    return ((x2 - x1) * (y2 - y1)) / ((a2 - a1) * (b2 - b1))


def update_clip(bbox_clip, start_idx, i, bbox_max):
    for j in range(start_idx, i):
        bbox_clip[j] = list(bbox_max)


ROI_THE = 0.5  # a frame whose box covers less than this share of the running clip box ends the clip (tools/util.py:214)


def frame_bbox(frame):
    """The padded person box of one pose frame: (x, x_max, y, y_max) (tools/util.py:183-189)."""
    mask = clean_mask(extract_mask_sdc(frame))
    y_, y_max_, x_, x_max_ = crop_img_sdc(frame, mask)
    x_, x_max_, y_, y_max_ = bbox_div2(x_, x_max_, y_, y_max_)
    return bbox_pad(x_, x_max_, y_, y_max_, frame)


def crop_human_clip_auto_context(pose_images, vid_images, bk_images, overlay=4):
    """Cuts the template into clips over which the person's box is stable and crops every frame of a clip to the clip's
    box; consecutive clips share `overlay` frames.  Returns (pose crops, video crops, background crops, per-frame clip
    boxes, context_list, bbox_clip_list) like the reference."""
    bbox_clip, bbox_perframe = [], []
    x, x_max, y, y_max = init_bbox()
    n_frame = len(pose_images)
    context_list, bbox_clip_list = [], []
    areas = np.zeros(n_frame)
    start_idx = 0

    def close_clip(end, box):
        if len(context_list) == 0:
            context_list.append(list(range(start_idx, end)))
        else:
            overlay_ = min(overlay, len(context_list[-1]))
            context_list.append(list(range(start_idx - overlay_, end)))
        bbox_clip_list.append(tuple(box))
        update_clip(bbox_clip, start_idx, end, box)

    for i in range(n_frame):
        frame = np.array(pose_images[i])
        x_, x_max_, y_, y_max_ = frame_bbox(frame)
        bbox_max_prev = (x, x_max, y, y_max)
        y, y_max, x, x_max = min(y, y_), max(y_max, y_max_), min(x, x_), max(x_max, x_max_)
        bbox_max_cur = (x, x_max, y, y_max)
        bbox_cur = [x_, x_max_, y_, y_max_]
        bbox_perframe.append(bbox_cur)
        bbox_clip.append(bbox_cur)
        areas[i] = (x_max_ - x_) * (y_max_ - y_) / 100
        area_max = (y_max - y) * (x_max - x) / 100
        ratios = areas[start_idx:i] / area_max if area_max != 0 else np.zeros(i - start_idx)
        if i == n_frame - 1:
            close_clip(i + 1, bbox_max_cur)
            start_idx = i + 1
        elif np.any(ratios < ROI_THE) and ratios.sum() != 0:
            close_clip(i, bbox_max_prev)
            x, x_max, y, y_max = bbox_cur
            start_idx = i

    frames_res, vid_res, bk_res = [], [], []
    for k, context in enumerate(context_list):
        for i in context:
            frame = np.array(pose_images[i])
            x, x_max, y, y_max = bbox_clip_list[k]
            if x >= x_max or y >= y_max:
                x, x_max, y, y_max = 0, frame.shape[1] - 1, 0, frame.shape[0] - 1
            frames_res.append(Image.fromarray(frame[y:y_max, x:x_max]))
            vid_res.append(Image.fromarray(np.array(vid_images[i])[y:y_max, x:x_max]))
            bk_res.append(Image.fromarray(np.array(bk_images[i])[y:y_max, x:x_max]))
    return frames_res, vid_res, bk_res, bbox_clip, context_list, bbox_clip_list


def prepare_clips(pose_crops, bk_crops):
    """run_edit.py:226-248: pad every cropped pose frame (black) and background frame (white) to a square multiple of 16.
    Returns (pose_list_context, vid_bk_list_context, clip_pad_list_context, clip_padv_list_context)."""
    pose_list, bk_list, pad_list, padv_list = [], [], [], []
    for pose_pil, bk_pil in zip(pose_crops, bk_crops):
        pose_image, _ = pad_img(np.array(pose_pil), color=[0, 0, 0])
        pose_list.append(Image.fromarray(pose_image))
        vid_bk, padding_v = pad_img(np.array(bk_pil), color=[255, 255, 255])
        pad_list.append([vid_bk.shape[0], vid_bk.shape[1]])
        padv_list.append(padding_v)
        bk_list.append(Image.fromarray(vid_bk))
    return pose_list, bk_list, pad_list, padv_list


def clip_masks(mask_list, context_list, bbox_clip_list, clip_pad_list, clip_padv_list, frame_size):
    """run_edit.py:283-284 for every generated frame: the edge mask of the clip's box (get_mask), area-resized to the
    un-padded crop size.  frame_size = (W, H) of the original frames.  Returns a list of float32 [h, w] arrays in
    generated-frame order — the `masks` argument of mimo_amd.edit.composite_clips."""
    class _Img:
        size = tuple(frame_size)
    out, video_idx = [], 0
    cache = {}
    for k, context in enumerate(context_list):
        bbox = bbox_clip_list[k]
        for _ in context:
            pad_h, pad_w = clip_pad_list[video_idx]
            top, bottom, left, right = clip_padv_list[video_idx]
            size = (pad_w - right - left, pad_h - bottom - top)  # res_image_pil.size after the crop (width, height)
            key = (k, size)
            if key not in cache:
                cache[key] = cvops.resize_area(np.asarray(get_mask(mask_list, bbox, _Img), np.float32), size)
            out.append(cache[key])
            video_idx += 1
    return out
