"""`MIMO.run` of the reference's video-editing entry point (run_edit.py:62-310), on the HIP path, end to end:

  reference image (cropped to the subject, padded white to a square)                    run_edit.py:163-169
  template frames at the target frame rate, time crop, frame cap                        run_edit.py:171-212, tools/util.py:462-479
  ROI-clip segmentation + per-frame padding                                             run_edit.py:217-239 (mimo_amd.template)
  Pose2VideoPipeline.__call__                                                           run_edit.py:241-251
  per-frame compositing (resize, un-pad, paste, edge mask, occluder, clip cross-fade)   run_edit.py:253-304 (mimo_amd.edit, on the device)

What is NOT here, and why: the TensorFlow matting graph `process_seg` (asset + runtime absent): the matting result is handed over
as an optional mask.  A template is either already decoded frames (`Template(...)`) or the reference's template DIRECTORY
(`Template.from_dir`: vid.mp4 / sdc.mp4 / bk.mp4 / occ.mp4 + config.json, run_edit.py:132-151) read through mimo_amd.video_io —
mp4 / mov / avi carrying Motion-JPEG, frame directories and animated images are decoded here, H.264 needs imageio + ffmpeg (absent
in this image; such a file raises, naming its codec).  `MIMO.run_paths` is the reference's `run(ref_img_path, template_path)` +
`imageio.mimsave` (run_edit.py:314-330) over the same containers.
`keep_frame_indices` and `time_crop_range` are the codec-free parts of `load_video_fixed_fps` / the time crop: which
decoded frames the reference keeps.
"""
import numpy as np
import torch
from PIL import Image

from . import edit as E
from . import template as T


def keep_frame_indices(n_frames, fps, target_fps=30, target_speed=1):
    """tools/util.py:462-479 `load_video_fixed_fps`: indices of the decoded frames kept when a video of `n_frames` at
    `fps` (rounded, as the reader's metadata is) is resampled to `target_fps`."""
    keep_ratio = target_speed * round(fps) / target_fps
    idx = np.arange(0, n_frames, keep_ratio).astype(int)
    return [int(i) for i in idx if i < n_frames]


def time_crop_range(target_fps, start_idx, end_idx, n_frames):
    """run_edit.py:194-198: the template's time crop is stored in 30-fps frame units."""
    s = max(0, int(target_fps * start_idx / 30))
    e = min(n_frames, int(target_fps * end_idx / 30))
    return s, e


class Template:
    """Decoded frames of a template directory (vid.mp4 / sdc.mp4 / bk.mp4 / occ.mp4 + config.json, run_edit.py:132-151):
    lists of PIL images (or uint8 arrays) at the videos' native `fps`; `occ` may be None; `bk` None = white backgrounds
    (tools/util.py `init_bk`)."""

    def __init__(self, vid, pose, bk=None, occ=None, fps=30, target_fps=None, time_crop=None):
        pil = lambda fr: None if fr is None else [f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f)) for f in fr]
        self.vid, self.pose, self.bk, self.occ = pil(vid), pil(pose), pil(bk), pil(occ)
        self.fps = fps
        self.target_fps = target_fps if target_fps is not None else fps
        self.time_crop = time_crop or {"start_idx": 0, "end_idx": 10 ** 9}

    @classmethod
    def from_dir(cls, template_path):
        """run_edit.py:132-151 `load_template` + the four `load_video_fixed_fps` readers' decode half (:171-190): vid / sdc / bk /
        occ as `<name>.mp4` (or .mov / .avi / .webp / a directory `<name>/` of stills, first that exists) and config.json
        (`fps`, `time_crop`).  The frame SELECTION stays in MIMO.select_frames (the reference resamples every video from its
        own rounded native rate to the template's fps)."""
        import json
        import os
        from . import video_io as V
        with open(os.path.join(template_path, "config.json")) as fh:
            cfg = json.load(fh)

        def find(name):
            for ext in (".mp4", ".mov", ".m4v", ".avi", ".webp", ".apng", ".gif", ""):
                p = os.path.join(template_path, name + ext)
                if os.path.exists(p):
                    return p
            return None

        def load(name, required=True):
            p = find(name)
            if p is None:
                if required:
                    raise FileNotFoundError(f"{template_path}: no {name}.mp4 (or .mov / .avi / frame directory)")
                return None, None
            return V.read_frames(p)

        vid, fps = load("vid")
        pose, fps_p = load("sdc")
        bk, _ = load("bk", required=False)
        occ, _ = load("occ", required=False)
        if round(fps_p) != round(fps):
            raise ValueError(f"{template_path}: vid and sdc have different frame rates ({fps} / {fps_p}); Template holds one `fps`")
        return cls(vid, pose, bk=bk, occ=occ, fps=fps, target_fps=cfg["fps"], time_crop=cfg.get("time_crop"))


class MIMO:
    """Same role as run_edit.py's `MIMO` with the models already built: `pipe` is a mimo_amd Pose2VideoPipeline,
    `mask_list` the 16 edge masks of `load_mask_list(assets/masks/alpha2.png)` (float32 arrays)."""

    def __init__(self, pipe, mask_list, width=784, height=784, steps=25, cfg=3.5, seed=42, max_frame_num=150):
        self.pipe, self.mask_list = pipe, mask_list
        self.width, self.height, self.steps, self.cfg = width, height, steps, cfg
        self.generator = torch.manual_seed(seed)
        self.max_frame_num = max_frame_num
        self.L = 0

    # -- run_edit.py:163-169 ------------------------------------------------------------------------------------
    @staticmethod
    def prepare_reference(ref_image, mask=None):
        """ref_image: PIL / uint8 RGB array; mask: the matting alpha as uint8 [H, W] (process_seg's second result) or None
        when the image is already the segmented subject.  -> PIL square image padded white."""
        src = np.asarray(ref_image.convert("RGB") if isinstance(ref_image, Image.Image) else ref_image)
        if mask is not None:
            src = T.crop_img(src, np.asarray(mask))
        src, _ = T.pad_img(src, [255, 255, 255])
        return Image.fromarray(src)

    # -- run_edit.py:171-212 ------------------------------------------------------------------------------------
    def select_frames(self, tpl):
        keep = lambda fr: None if fr is None else [fr[i] for i in keep_frame_indices(len(fr), tpl.fps, tpl.target_fps)]
        vid, pose, occ = keep(tpl.vid), keep(tpl.pose), keep(tpl.occ)
        if tpl.bk is None:
            tw, th = vid[0].size
            bk = [Image.fromarray(np.full((th, tw, 3), 255, np.uint8)) for _ in vid]
        else:
            bk = keep(tpl.bk)
        s, e = time_crop_range(tpl.target_fps, tpl.time_crop["start_idx"], tpl.time_crop["end_idx"], len(pose))
        cut = lambda fr: None if fr is None else fr[s:e][:self.max_frame_num]
        return cut(vid), cut(pose), cut(bk), cut(occ)

    # -- run_edit.py:153-306 ------------------------------------------------------------------------------------
    def run(self, ref_image, tpl, ref_mask=None, overlay=4, return_device=False):
        """-> (res_images, target_fps): uint8 [H, W, 3] frames (a list of arrays as in the reference, or the device tensor
        [L, H, W, 3] with return_device=True).  Frames no clip covers stay zero (the reference leaves None there)."""
        ref_image_pil = self.prepare_reference(ref_image, ref_mask)
        vid_images, pose_images, bk_images, occ_mask_images = self.select_frames(tpl)
        self.L = len(pose_images)
        bk_images_ori, vid_images_ori = list(bk_images), list(vid_images)
        pose_c, vid_c, bk_c, bbox_clip, context_list, bbox_clip_list = T.crop_human_clip_auto_context(
            pose_images, vid_images, bk_images, overlay)
        pose_list, bk_list, clip_pad_list, clip_padv_list = T.prepare_clips(pose_c, bk_c)
        out = self.pipe(ref_image_pil, pose_list, bk_list, self.width, self.height, len(pose_list), self.steps, self.cfg,
                        generator=self.generator, output_device=True)
        video = out.videos[0]
        masks = T.clip_masks(self.mask_list, context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk_images_ori[0].size)
        res = E.composite_clips(video.contiguous(), context_list, bbox_clip_list, clip_pad_list, clip_padv_list, bk_images_ori,
                                vid_images_ori, occ_mask_images, masks, overlay=overlay, L=self.L)
        self.last = dict(context_list=context_list, bbox_clip_list=bbox_clip_list, clip_pad_list=clip_pad_list,
                         clip_padv_list=clip_padv_list, masks=masks, pose_list=pose_list, bk_list=bk_list,
                         ref_image=ref_image_pil, frames=(vid_images_ori, bk_images_ori, occ_mask_images), video=video)
        if return_device:
            return res, tpl.target_fps
        host = res.cpu().numpy()
        return [host[i] for i in range(host.shape[0])], tpl.target_fps

    def run_paths(self, ref_img_path, template_path, outpath, ref_mask=None, **save_kw):
        """run_edit.py:314-330 (`main`): reference image file + template directory in, video file out (`imageio.mimsave(outpath,
        res, fps=target_fps)` there; mimo_amd.video_io.save_video here: .mp4 = imageio + ffmpeg when present, else Motion-JPEG mp4)."""
        from . import video_io as V
        res, fps = self.run(Image.open(ref_img_path).convert("RGB"), Template.from_dir(template_path), ref_mask=ref_mask)
        return V.save_video(res, outpath, fps, **save_kw)
