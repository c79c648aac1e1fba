"""Frame I/O either side of `MIMO.run` — the asset half of SURVEY 8(f) rank 3 — without a video codec.

The reference reads its templates with `imageio.get_reader(...)` (`tools/util.py:462-479` `load_video_fixed_fps`) and writes results
with `imageio.mimsave(outpath, res, fps=target_fps, ...)` (`run_animate.py:248`, `run_edit.py:328`), i.e. mp4 through ffmpeg.  This
image has neither imageio nor a codec, so the same two functions are provided over the containers Pillow handles bit-exactly —

  a directory of numbered still images (png / jpg / webp ...; the frame rate comes from `fps.txt` / `config.json` beside them or
  the `fps` argument), animated WebP (lossless), APNG, GIF, AVI: Motion-JPEG (the one video codec this image does have —
  Pillow's JPEG — in the RIFF container every player, ffmpeg and OpenCV read) or uncompressed 24-bit frames (lossless), and
  **MP4 / MOV (ISO base media file format) carrying Motion-JPEG** (round 6): the container the reference's templates ship in
  (`vid.mp4`, `bk.mp4`, `mask.mp4`, `sdc.mp4`: `run_edit.py:132-239`) and its results are written to, muxed and demuxed here
  (ftyp / moov / mdat with a `mp4v` sample entry whose esds declares object type 0x6C = JPEG, exactly what
  `ffmpeg -c:v mjpeg out.mp4` writes; the QuickTime `jpeg` / `mjpa` entries are read too)

— and over H.264 / HEVC mp4 whenever `imageio` can be imported (the branch the reference itself takes; no such decoder exists in
this image: a foreign codec raises with the sample entry's name).  The frame
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

_STILLS = (".png", ".jpg", ".jpeg", ".webp", ".bmp")   # compared against the lower-cased file extension
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
    if codec not in ("mjpeg", "raw"):
        raise ValueError(f"AVI codec {codec!r}: 'mjpeg' and 'raw' are available")
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


# ---- MP4 / MOV (ISO/IEC 14496-12) with Motion-JPEG samples: one video track, faststart layout (ftyp, moov, mdat) ----
def _box(kind, payload):
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def _full(kind, version, flags, payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag, payload):
    """MPEG-4 descriptor: tag, size in the 4-byte 0x80-continued form ffmpeg writes, payload."""
    n = len(payload)
    return bytes([tag, 0x80 | (n >> 21) & 0x7f, 0x80 | (n >> 14) & 0x7f, 0x80 | (n >> 7) & 0x7f, n & 0x7f]) + payload


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def write_mp4(frames, path, fps, quality=95):
    """frames: RGB PIL images of one size -> `path` (.mp4 / .m4v / .mov): JPEG per frame (`quality`, 4:4:4) as the samples of one
    video track.  Sample entry `mp4v` + esds with objectTypeIndication 0x6C (ISO/IEC 10918-1 JPEG, the registered way to carry
    JPEG in MP4; ffmpeg, VLC and QuickTime play it); constant sample duration, every sample a sync sample (no stss)."""
    w, h = frames[0].size
    num, den = (int(round(fps * 1000)), 1000) if abs(fps - round(fps)) > 1e-9 else (int(round(fps)) * 1000, 1000)
    data = []
    for f in frames:
        assert f.size == (w, h), "all frames of a video have one size"
        buf = io.BytesIO()
        f.convert("RGB").save(buf, format="JPEG", quality=quality, subsampling=0)
        data.append(buf.getvalue())
    n = len(data)
    dur = n * den
    total = sum(len(d) for d in data)
    biggest = max(len(d) for d in data)
    brate = int(total * 8 * num / max(dur, 1))
    dec_cfg = _descr(4, bytes([0x6C, 0x11]) + struct.pack(">I", biggest)[1:] + struct.pack(">II", brate, brate))   # visual stream, JPEG
    esds = _full(b"esds", 0, 0, _descr(3, struct.pack(">HB", 1, 0) + dec_cfg + _descr(6, b"\x02")))
    entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HHIIIH", w, h, 0x480000, 0x480000, 0, 1) +
             bytes([10]) + b"mimo_amd  "[:10].ljust(31, b"\0") + struct.pack(">Hh", 24, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", entry))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, den))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))                        # ONE chunk holds every sample
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(d)) for d in data))

    def moov(chunk_offset):
        big = chunk_offset + total >= 1 << 32
        stco = _full(b"co64" if big else b"stco", 0, 0, struct.pack(">IQ" if big else ">II", 1, chunk_offset))
        stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
        dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
        minf = _box(b"minf", _full(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0)) + dinf + stbl)
        mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, num, dur, 0x55C4, 0))
        hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4sIII", 0, b"vide", 0, 0, 0) + b"VideoHandler\0")
        tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur) + b"\0" * 8 + struct.pack(">hhhH", 0, 0, 0, 0) + _MATRIX +
                     struct.pack(">II", w << 16, h << 16))
        trak = _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))
        mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, num, dur, 0x10000, 0x100) + b"\0" * 10 + _MATRIX + b"\0" * 24 +
                     struct.pack(">I", 2))
        return _box(b"moov", mvhd + trak)

    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2mp41")
    head = 16 if total + 8 >= 1 << 32 else 8
    size = len(moov(0))
    mv = moov(len(ftyp) + size + head)
    if len(mv) != size:                      # the chunk offset crossed 4 GiB: co64 is four bytes longer
        mv = moov(len(ftyp) + len(mv) + head)
    with open(path, "wb") as fh:
        fh.write(ftyp + mv)
        fh.write(struct.pack(">I4sQ", 1, b"mdat", total + 16) if head == 16 else struct.pack(">I4s", total + 8, b"mdat"))
        for d in data:
            fh.write(d)
    return path


def _esds_object_type(entry):
    """objectTypeIndication of the DecoderConfigDescriptor inside a sample entry's esds box, or -1."""
    k = entry.find(b"esds")
    if k < 0:
        return -1
    p = k + 8                                      # behind the box type + version / flags

    def descr(p):
        tag, n = entry[p], 0
        p += 1
        while True:
            b = entry[p]
            p += 1
            n = (n << 7) | (b & 0x7f)
            if not b & 0x80:
                return tag, n, p

    try:
        tag, n, p = descr(p)
        if tag != 3:
            return -1
        flags = entry[p + 2]
        p += 3 + (2 if flags & 0x80 else 0) + (2 if flags & 0x20 else 0)
        if flags & 0x40:
            p += 1 + entry[p]
        tag, n, p = descr(p)
        return entry[p] if tag == 4 else -1
    except IndexError:
        return -1


_CONTAINERS = {b"moov", b"trak", b"mdia", b"minf", b"stbl", b"edts", b"dinf", b"udta"}


def _mp4_boxes(raw, lo, hi, out, path=()):
    p = lo
    while p + 8 <= hi:
        size, kind = struct.unpack(">I4s", raw[p:p + 8])
        body = p + 8
        if size == 1:
            size = struct.unpack(">Q", raw[p + 8:p + 16])[0]
            body = p + 16
        elif size == 0:
            size = hi - p
        if size < 8 or p + size > hi:
            break
        out.append((path + (kind,), body, p + size))
        if kind in _CONTAINERS:
            _mp4_boxes(raw, body, p + size, out, path + (kind,))
        p += size
    return out


def read_mp4(path):
    """(RGB PIL frames, frame rate) of the first video track of an MP4 / MOV file whose samples are JPEG images: `mp4v` with
    esds object type 0x6C (what write_mp4 and `ffmpeg -c:v mjpeg x.mp4` produce), or the QuickTime entries `jpeg` / `mjpa` /
    `MJPG`.  Chunk layout is general (stsc runs, stco / co64, per-sample or constant stsz).  Any other sample entry (avc1, hvc1,
    av01 ...) raises RuntimeError naming it: there is no decoder for it in this environment."""
    with open(path, "rb") as fh:
        raw = memoryview(fh.read())
    boxes = _mp4_boxes(raw, 0, len(raw), [])
    if not boxes or boxes[0][0][-1] not in (b"ftyp", b"moov", b"mdat", b"free", b"wide", b"skip"):
        raise ValueError(f"{path}: not an ISO base media (mp4 / mov) file")
    traks = [b for b in boxes if b[0][-1] == b"trak"]
    for _, tlo, thi in traks:
        inside = {b[0][-1]: (b[1], b[2]) for b in boxes if tlo <= b[1] and b[2] <= thi}
        if b"hdlr" not in inside or bytes(raw[inside[b"hdlr"][0] + 8:inside[b"hdlr"][0] + 12]) != b"vide":
            continue
        lo, _ = inside[b"mdhd"]
        ver = raw[lo]
        timescale = struct.unpack(">I", raw[lo + (20 if ver == 1 else 12):lo + (24 if ver == 1 else 16)])[0]
        lo, hi = inside[b"stsd"]
        esize, fmt = struct.unpack(">I4s", raw[lo + 8:lo + 16])
        entry = bytes(raw[lo + 8:lo + 8 + esize])
        fmt = bytes(fmt)
        ok = fmt in (b"jpeg", b"mjpa", b"MJPG", b"mjpg", b"AVDJ")
        if fmt == b"mp4v":                 # MPEG-4 visual sample entry: the esds names the codec (0x6C = JPEG, 0x20 = MPEG-4 part 2)
            ok = _esds_object_type(entry) == 0x6C
        if not ok:
            raise RuntimeError(f"{path}: video sample entry {fmt!r} — no decoder for it in this environment "
                               "(Motion-JPEG in mp4 / mov / avi is read here; H.264 / HEVC need imageio + ffmpeg)")
        lo, hi = inside[b"stts"]
        cnt = struct.unpack(">I", raw[lo + 4:lo + 8])[0]
        pairs = [struct.unpack(">II", raw[lo + 8 + 8 * i:lo + 16 + 8 * i]) for i in range(cnt)]
        nsamp_t, ticks = sum(c for c, _ in pairs), sum(c * d for c, d in pairs)
        lo, hi = inside[b"stsz"]
        const, n = struct.unpack(">II", raw[lo + 4:lo + 12])
        sizes = [const] * n if const else list(struct.unpack(f">{n}I", raw[lo + 12:lo + 12 + 4 * n]))
        if b"co64" in inside:
            lo, hi = inside[b"co64"]
            nc = struct.unpack(">I", raw[lo + 4:lo + 8])[0]
            chunks = list(struct.unpack(f">{nc}Q", raw[lo + 8:lo + 8 + 8 * nc]))
        else:
            lo, hi = inside[b"stco"]
            nc = struct.unpack(">I", raw[lo + 4:lo + 8])[0]
            chunks = list(struct.unpack(f">{nc}I", raw[lo + 8:lo + 8 + 4 * nc]))
        lo, hi = inside[b"stsc"]
        nr = struct.unpack(">I", raw[lo + 4:lo + 8])[0]
        runs = [struct.unpack(">III", raw[lo + 8 + 12 * i:lo + 20 + 12 * i]) for i in range(nr)]
        frames, s = [], 0
        for ci, off in enumerate(chunks):                              # samples per chunk: the last run whose first_chunk <= ci + 1
            per = [r[1] for r in runs if r[0] <= ci + 1][-1]
            for _ in range(per):
                if s >= n:
                    break
                d = bytes(raw[off:off + sizes[s]])
                if fmt == b"mjpa" and d[:2] != b"\xff\xd8":
                    raise RuntimeError(f"{path}: unsupported Motion-JPEG-A field layout")
                frames.append(Image.open(io.BytesIO(d)).convert("RGB"))
                off += sizes[s]
                s += 1
        rate = timescale * nsamp_t / ticks if ticks else 30.0
        return frames, rate
    raise ValueError(f"{path}: no video track")


_MP4 = (".mp4", ".m4v", ".mov")


def read_frames(path, fps=None):
    """(frames as RGB PIL images, frame rate).  `path`: a directory of stills (sorted by name), an AVI / MP4 / MOV with Motion-JPEG
    (or uncompressed AVI) video, an animated image, or — with imageio installed — anything imageio reads.  The returned rate is
    the caller's `fps` when given, else the container's own (fps.txt / config.json beside a directory of stills; default 30)."""
    if os.path.isdir(path):
        names = sorted(n for n in os.listdir(path) if n.lower().endswith(_STILLS))
        if not names:
            raise FileNotFoundError(f"no frames in {path}")
        # one precedence for every container: the caller's explicit `fps` over the container's own metadata
        return [Image.open(os.path.join(path, n)).convert("RGB") for n in names], (_dir_fps(path, 30.0) if fps is None else fps)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".avi":
        frames, native = read_avi(path)
        return frames, (native if fps is None else fps)
    if ext in _ANIMATED:
        im = Image.open(path)
        frames = [f.convert("RGB") for f in ImageSequence.Iterator(im)]
        dur = im.info.get("duration", 0) or 0
        return frames, (fps if fps is not None else (1000.0 / dur if dur > 0 else 30.0))
    foreign = None
    if ext in _MP4:
        try:
            frames, native = read_mp4(path)
            return frames, (native if fps is None else fps)
        except RuntimeError as e:       # a codec Pillow cannot decode: imageio + ffmpeg if they exist (the reference's own route)
            foreign = e
    try:
        import imageio
    except ImportError as e:
        if foreign is not None:
            raise foreign
        raise RuntimeError(f"{path}: no video codec in this environment (imageio is not installed); hand the template over as a "
                           "directory of frames, a Motion-JPEG mp4 / mov / avi, an uncompressed AVI or an animated WebP / APNG / GIF") from e
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
    uncompressed, lossless), .mp4 / .m4v / .mov (imageio + ffmpeg when installed, else — or with codec='mjpeg!' — the Motion-JPEG
    mp4 of write_mp4), .webp (lossless animation), .png / .apng, .gif; anything else goes to imageio when it is installed."""
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
        return write_avi(pil, outpath, fps, codec="mjpeg" if codec == "mjpeg!" else codec, quality=quality)
    if ext in _MP4:
        # imageio + ffmpeg (H.264, what the reference's imageio.mimsave writes) when they exist and the caller did not ask for
        # Motion-JPEG by name; otherwise the Motion-JPEG mp4 muxed here — a real .mp4 either way
        have_imageio = False
        if codec != "mjpeg!":
            try:
                import imageio  # noqa: F401
                have_imageio = True
            except ImportError:
                pass
        if not have_imageio:
            if codec not in ("mjpeg", "mjpeg!"):
                raise ValueError(f"{outpath}: codec {codec!r} is not available for mp4 here (Motion-JPEG is)")
            return write_mp4(pil, outpath, fps, quality=quality)
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
