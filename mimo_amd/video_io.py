"""Frame I/O either side of `MIMO.run` — the asset half of SURVEY 8(f) rank 3 — without a video codec.

The reference reads its templates with `imageio.get_reader(...)` (`tools/util.py:462-479` `load_video_fixed_fps`) and writes results
with `imageio.mimsave(outpath, res, fps=target_fps, ...)` (`run_animate.py:248`, `run_edit.py:328`), i.e. mp4 through ffmpeg.  This
image has neither imageio nor a codec, so the same two functions are provided over the containers Pillow handles bit-exactly —

  a directory of numbered still images (png / jpg / webp ...; the frame rate comes from `fps.txt` / `config.json` beside them or
  the `fps` argument), animated WebP (lossless), APNG, GIF, and AVI: Motion-JPEG (the one video codec this image does have —
  Pillow's JPEG — in the RIFF container every player, ffmpeg and OpenCV read) or uncompressed 24-bit frames (lossless)

— and over mp4 whenever `imageio` can be imported (the branch the reference itself takes; never exercised here).  The frame
SELECTION is the reference's arithmetic exactly (`run_edit.keep_frame_indices`: metadata fps rounded, `np.arange(0, n, ratio).astype(int)`),
pinned against the reference's own function in tests/test_host_cpu.py.
"""
import io
import json
import os
import struct

import numpy as np
from PIL import Image, ImageSequence

from .run_edit import keep_frame_indices

_STILLS = (".png", ".jpg", ".jpeg", ".webp", ".bmp", ".JPG")
_ANIMATED = (".webp", ".gif", ".png", ".apng")


def _dir_fps(path, default):
    for name, get in (("fps.txt", lambda t: float(t.strip())), ("config.json", lambda t: float(json.loads(t)["fps"]))):
        f = os.path.join(path, name)
        if os.path.exists(f):
            return get(open(f).read())
    return default


# ---- AVI (RIFF) with Motion-JPEG or uncompressed 24-bit DIB frames: one video stream, an idx1 index, < 2 GiB (no OpenDML) ----
def _chunk(fourcc, payload):
    return fourcc + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")


def write_avi(frames, path, fps, codec="mjpeg", quality=95):
    """frames: RGB PIL images of one size.  codec 'mjpeg' (JPEG per frame, `quality`) or 'raw' (bottom-up BGR rows: lossless)."""
    w, h = frames[0].size
    num, den = (int(round(fps * 1000)), 1000) if abs(fps - round(fps)) > 1e-9 else (int(round(fps)), 1)
    data = []
    for f in frames:
        assert f.size == (w, h), "all frames of a video have one size"
        if codec == "mjpeg":
            buf = io.BytesIO()
            f.convert("RGB").save(buf, format="JPEG", quality=quality, subsampling=0)
            data.append(buf.getvalue())
        else:
            a = np.asarray(f.convert("RGB"))[::-1, :, ::-1].reshape(h, w * 3)   # bottom-up rows of BGR, padded to 4 bytes
            data.append(np.concatenate([a, np.zeros((h, (-w * 3) % 4), np.uint8)], axis=1).tobytes())
    fourcc, ckid = (b"MJPG", b"00dc") if codec == "mjpeg" else (b"\0\0\0\0", b"00db")
    biggest = max(len(d) for d in data)
    avih = struct.pack("<14I", int(round(1e6 * den / num)), int(biggest * num / den), 0, 0x10, len(data), 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + (fourcc if codec == "mjpeg" else b"DIB ") + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, den, num, 0, len(data), biggest,
                                                                                 0xffffffff, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, fourcc, w * h * 3, 0, 0, 0, 0)
    hdrl = b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf))
    movi, idx, off = b"movi", b"", 4                                               # idx1 offsets count from the 'movi' fourcc
    for d in data:
        idx += ckid + struct.pack("<III", 0x10, off, len(d))
        c = _chunk(ckid, d)
        movi += c
        off += len(c)
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi) + _chunk(b"idx1", idx)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


def read_avi(path):
    """(RGB PIL frames, frame rate) of an AVI whose video stream is Motion-JPEG (frames that carry their Huffman tables, as ffmpeg,
    OpenCV and write_avi make them) or uncompressed 24-bit."""
    raw = open(path, "rb").read()
    if raw[:4] != b"RIFF" or raw[8:12] != b"AVI ":
        raise ValueError(f"{path}: not a RIFF AVI file")
    info = {}
    frames = []

    def walk(lo, hi):
        p = lo
        while p + 8 <= hi:
            cc, n = raw[p:p + 4], struct.unpack("<I", raw[p + 4:p + 8])[0]
            body = p + 8
            if cc == b"LIST":
                walk(body + 4, body + n)
            elif cc == b"strh" and raw[body:body + 4] == b"vids" and "rate" not in info:
                scale, rate = struct.unpack("<II", raw[body + 20:body + 28])
                info["rate"] = rate / max(scale, 1)
            elif cc == b"strf" and "w" not in info:
                _, w, h, _, bits, comp = struct.unpack("<IiiHH4s", raw[body:body + 20])
                info.update(w=w, h=h, bits=bits, comp=comp)
            elif cc[2:] in (b"dc", b"db") and cc[:2].isdigit() and n:
                frames.append(raw[body:body + n])
            p = body + n + (n & 1)

    walk(12, len(raw))
    if "w" not in info or not frames:
        raise ValueError(f"{path}: no video stream")
    comp = info["comp"]
    if comp in (b"MJPG", b"mjpg", b"MJPEG"[:4], b"jpeg", b"JPEG"):
        out = [Image.open(io.BytesIO(d)).convert("RGB") for d in frames]
    elif comp in (b"\0\0\0\0", b"DIB ", b"RGB ") and info["bits"] == 24:
        w, h = info["w"], abs(info["h"])
        stride = (w * 3 + 3) & ~3
        out = []
        for d in frames:
            a = np.frombuffer(d, np.uint8, count=stride * h).reshape(h, stride)[:, :w * 3].reshape(h, w, 3)[:, :, ::-1]
            out.append(Image.fromarray(np.ascontiguousarray(a[::-1] if info["h"] > 0 else a)))
    else:
        raise RuntimeError(f"{path}: AVI video codec {comp!r} is not available here (Motion-JPEG and uncompressed 24-bit are)")
    return out, info.get("rate", 30.0)


def read_frames(path, fps=None):
    """(frames as RGB PIL images, native frame rate).  `path`: a directory of stills (sorted by name), an animated image, or —
    with imageio installed — anything imageio reads."""
    if os.path.isdir(path):
        names = sorted(n for n in os.listdir(path) if n.endswith(_STILLS))
        if not names:
            raise FileNotFoundError(f"no frames in {path}")
        return [Image.open(os.path.join(path, n)).convert("RGB") for n in names], _dir_fps(path, 30.0 if fps is None else fps)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".avi":
        frames, native = read_avi(path)
        return frames, (native if fps is None else fps)
    if ext in _ANIMATED:
        im = Image.open(path)
        frames = [f.convert("RGB") for f in ImageSequence.Iterator(im)]
        dur = im.info.get("duration", 0) or 0
        return frames, (fps if fps is not None else (1000.0 / dur if dur > 0 else 30.0))
    try:
        import imageio
    except ImportError as e:
        raise RuntimeError(f"{path}: no video codec in this environment (imageio is not installed); hand the template over as a "
                           "directory of frames, a Motion-JPEG / uncompressed AVI or an animated WebP / APNG / GIF") from e
    reader = imageio.get_reader(path)
    native = reader.get_meta_data()["fps"]
    frames = [Image.fromarray(reader.get_data(i)) for i in range(reader.count_frames())]
    reader.close()
    return frames, native


def load_video_fixed_fps(vid_path, target_fps=30, target_speed=1, fps=None):
    """tools/util.py:462-479: the frames of `vid_path` resampled to `target_fps` (list of PIL images)."""
    frames, native = read_frames(vid_path, fps)
    return [frames[i] for i in keep_frame_indices(len(frames), native, target_fps, target_speed)]


def save_video(frames, outpath, fps, codec="mjpeg", quality=95):
    """The role of `imageio.mimsave(outpath, res, fps=target_fps)` (run_animate.py:248): frames = uint8 [H, W, 3] arrays or PIL
    images.  By extension: a directory (no extension; numbered PNGs + fps.txt), .avi (Motion-JPEG at `quality`, or codec='raw':
    uncompressed, lossless), .webp (lossless animation), .png / .apng, .gif; anything else goes to imageio when it is installed."""
    pil = [f if isinstance(f, Image.Image) else Image.fromarray(np.asarray(f)) for f in frames]
    ext = os.path.splitext(outpath)[1].lower()
    if ext == "":
        os.makedirs(outpath, exist_ok=True)
        for i, f in enumerate(pil):
            f.save(os.path.join(outpath, f"{i:05d}.png"))
        with open(os.path.join(outpath, "fps.txt"), "w") as fh:
            fh.write(f"{fps}\n")
        return outpath
    if ext == ".avi":
        return write_avi(pil, outpath, fps, codec=codec, quality=quality)
    dur = int(round(1000.0 / fps))
    if ext == ".webp":
        pil[0].save(outpath, save_all=True, append_images=pil[1:], duration=dur, loop=0, lossless=True, quality=100, method=4)
    elif ext in (".png", ".apng"):
        pil[0].save(outpath, save_all=True, append_images=pil[1:], duration=dur, loop=0)
    elif ext == ".gif":
        pil[0].save(outpath, save_all=True, append_images=pil[1:], duration=dur, loop=0)
    else:
        try:
            import imageio
        except ImportError as e:
            raise RuntimeError(f"{outpath}: no video codec in this environment; use a directory, .avi, .webp, .apng or .gif") from e
        imageio.mimsave(outpath, [np.asarray(f) for f in pil], fps=fps, quality=8, macro_block_size=1)
    return outpath
