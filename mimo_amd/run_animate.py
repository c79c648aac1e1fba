"""`MIMO.run` of the reference's character-animation entry point (run_animate.py:153-229), on the HIP path:

  reference image: matting mask -> crop_img -> pad_img white                            run_animate.py:162-168
  driving frames (the template's sdc.mp4) at 30 fps, white backgrounds, frame cap        run_animate.py:170-191, tools/util.py:462-479
  ONE human-centred crop for the whole clip (crop_human)                                run_animate.py:193-194, tools/util.py:71-110
  per-frame padding: pose black, background white                                       run_animate.py:196-206
  Pose2VideoPipeline.__call__                                                           run_animate.py:208-219
  frames back as uint8 ((image * 255).astype(uint8): truncation, as the reference)      run_animate.py:221-227

Not here, as in mimo_amd.run_edit: the TensorFlow matting graph `process_seg` (the matting result is an optional mask).  The
driving video is decoded frames, or — `MIMO.run_paths`, the reference's `run(ref_img_path, template_path)` + `imageio.mimsave`
(run_animate.py:153-160, 231-249) — the template directory's `sdc.mp4` read through mimo_amd.video_io (Motion-JPEG mp4 / mov /
avi, frame directories; H.264 needs imageio + ffmpeg).
"""
import numpy as np
import torch
from PIL import Image

from . import template as T
from .run_edit import keep_frame_indices


class MIMO:
    """Same role as run_animate.py's `MIMO` with the models already built: `pipe` is a mimo_amd Pose2VideoPipeline."""

    def __init__(self, pipe, width=784, height=784, steps=25, cfg=3.5, seed=42, max_frame_num=150):
        self.pipe = pipe
        self.width, self.height, self.steps, self.cfg = width, height, steps, cfg
        self.generator = torch.manual_seed(seed)
        self.max_frame_num = max_frame_num
        self.L = 0

    @staticmethod
    def prepare_reference(ref_image, mask=None):
        """run_animate.py:162-168.  ref_image: PIL / uint8 RGB array; mask: the matting alpha (uint8 [H, W]) or None when the
        image already is the segmented subject on white.  -> PIL square image padded white."""
        src = np.asarray(ref_image.convert("RGB") if isinstance(ref_image, Image.Image) else ref_image)
        if mask is not None:
            src = T.crop_img(src, np.asarray(mask))
        src, _ = T.pad_img(src, [255, 255, 255])
        return Image.fromarray(src)

    def prepare_frames(self, pose_frames, fps=30, bk_frames=None):
        """run_animate.py:170-206 -> (pose_list, vid_bk_list): padded PIL frames as handed to the pipeline."""
        target_fps = 30
        pil = lambda fr: [f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f)) for f in fr]
        keep = lambda fr: [fr[i] for i in keep_frame_indices(len(fr), fps, target_fps)]
        pose_images = keep(pil(pose_frames))
        vid_images = list(pose_images)                 # video_path = pose_video_path = sdc.mp4 (run_animate.py:159-160)
        if bk_frames is None:
            tw, th = vid_images[0].size
            # (as the reference calls it: init_bk(n_frame, tw, th) with the signature (n_frame, h, w) — for a non-square
            # driving video the white frames are transposed; they are white everywhere, only the crop below sees the shape)
            bk_images = T.init_bk(len(vid_images), tw, th)
        else:
            bk_images = keep(pil(bk_frames))
        pose_images, vid_images, bk_images = (fr[:self.max_frame_num] for fr in (pose_images, vid_images, bk_images))
        self.L = len(pose_images)
        pose_images, vid_images, bk_images = T.crop_human(pose_images, vid_images, bk_images)
        pose_list = [Image.fromarray(T.pad_img(np.array(p), color=[0, 0, 0])[0]) for p in pose_images]
        vid_bk_list = [Image.fromarray(T.pad_img(np.array(b), color=[255, 255, 255])[0]) for b in bk_images]
        return pose_list, vid_bk_list

    def run(self, ref_image, pose_frames, fps=30, ref_mask=None, bk_frames=None, return_device=False):
        """-> (res_images, target_fps = 30): uint8 [H, W, 3] frames (a list of arrays; PIL in the reference), or the device
        tensor [L, H, W, 3] with return_device=True."""
        ref_image_pil = self.prepare_reference(ref_image, ref_mask)
        pose_list, vid_bk_list = self.prepare_frames(pose_frames, fps, bk_frames)
        out = self.pipe(ref_image_pil, pose_list, vid_bk_list, self.width, self.height, len(pose_list), self.steps, self.cfg,
                        generator=self.generator, output_device=True)
        video = out.videos[0]                          # [3, L, H, W] float32 in [0, 1]
        res = (video.permute(1, 2, 3, 0) * 255).to(torch.uint8).contiguous()
        self.last = dict(ref_image=ref_image_pil, pose_list=pose_list, bk_list=vid_bk_list, video=video)
        if return_device:
            return res, 30
        host = res.cpu().numpy()
        return [host[i] for i in range(host.shape[0])], 30

    def run_paths(self, ref_img_path, template_path, outpath, ref_mask=None, **save_kw):
        """run_animate.py:153-160 + 231-249 (`main`): reference image file + template directory (its `sdc.mp4`, or .mov / .avi / a
        frame directory of that name) in, video file out at 30 fps."""
        import os
        from . import video_io as V
        src = next((os.path.join(template_path, "sdc" + e) for e in (".mp4", ".mov", ".m4v", ".avi", ".webp", "")
                    if os.path.exists(os.path.join(template_path, "sdc" + e))), None)
        if src is None:
            raise FileNotFoundError(f"{template_path}: no sdc.mp4 (or .mov / .avi / frame directory)")
        frames, fps = V.read_frames(src)
        res, target_fps = self.run(Image.open(ref_img_path).convert("RGB"), frames, fps=fps, ref_mask=ref_mask)
        return V.save_video(res, outpath, target_fps, **save_kw)
