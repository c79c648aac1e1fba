"""The OpenCV primitives the reference's template pre/post-processing calls (tools/util.py, run_edit.py:284), on NumPy.

cv2 is not part of this image (no network), so these follow OpenCV 4.x's documented algorithms:
  rgb2gray            cv2.cvtColor(img, cv2.COLOR_RGB2GRAY), 8-bit (OpenCV 4.x: 15-bit weights): (R 9798 + G 19235 + B 3735 + 2^14) >> 15
  morphology_rect     cv2.morphologyEx(mask, MORPH_CLOSE | MORPH_OPEN, getStructuringElement(MORPH_RECT, (k, k))):
                      anchor = k // 2, the border never contributes (morphologyDefaultBorderValue)
  bounding_rect       cv2.boundingRect(mask) of the non-zero pixels -> (x, y, w, h); (0, 0, 0, 0) when empty
  copy_make_border    cv2.copyMakeBorder(img, t, b, l, r, BORDER_CONSTANT, value=color)
  resize_area         cv2.resize(src, (w, h), interpolation=cv2.INTER_AREA) for float32 / uint8 images: the integer-ratio
                      fast path, the fractional decimation tables (computeResizeAreaTab) and — when an axis is enlarged —
                      the bilinear form INTER_AREA falls back to, with its area-style coefficients
They are integer / index logic except resize_area, whose float32 accumulation order follows resizeArea_ / the linear
resizer (columns first, then rows); without cv2 here that last bit is pinned by construction only (integer ratios equal
block means exactly, constant images stay constant, PIL's BOX filter agrees to float rounding) — tests/test_host_cpu.py."""
import math

import numpy as np


def rgb2gray(img):
    """uint8 [H, W, 3] RGB -> uint8 [H, W]: OpenCV 4.x's 8-bit COLOR_RGB2GRAY (color_yuv.simd.hpp: RY15 = 9798, GY15 = 19235,
    BY15 = 3735, gray_shift = 15, rounded).  OpenCV 3.x used the 14-bit weights 4899 / 9617 / 1868: the two differ by one
    grey level on ~0.1 % of the colours, which matters only next to the `gray > 10` threshold of extract_mask_sdc.  The
    reference pins no OpenCV version (it arrives through controlnet-aux): the 4.x arithmetic is the one installed today."""
    a = img.astype(np.int32)
    return ((a[..., 0] * 9798 + a[..., 1] * 19235 + a[..., 2] * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def _shifted_reduce(m, k, fn, fill):
    """fn over the k x k window whose anchor (k // 2, k // 2) sits on the pixel; outside the image = `fill`."""
    H, W = m.shape
    a = k // 2
    p = np.full((H + k - 1, W + k - 1), fill, m.dtype)
    p[a:a + H, a:a + W] = m
    out = None
    for dy in range(k):
        for dx in range(k):
            v = p[dy:dy + H, dx:dx + W]
            out = v.copy() if out is None else fn(out, v)
    return out


def dilate_rect(m, k):
    return _shifted_reduce(m, k, np.maximum, np.iinfo(m.dtype).min)


def erode_rect(m, k):
    return _shifted_reduce(m, k, np.minimum, np.iinfo(m.dtype).max)


def morphology_rect(m, op, k):
    """op 'close' = erode(dilate(m)), 'open' = dilate(erode(m)) with a k x k rectangle."""
    if op == "close":
        return erode_rect(dilate_rect(m, k), k)
    if op == "open":
        return dilate_rect(erode_rect(m, k), k)
    raise ValueError(op)


def bounding_rect(mask):
    ys, xs = np.nonzero(mask)
    if ys.size == 0:
        return 0, 0, 0, 0
    x0, x1, y0, y1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
    return x0, y0, x1 - x0 + 1, y1 - y0 + 1


def copy_make_border(img, top, bottom, left, right, color):
    H, W = img.shape[:2]
    out = np.empty((H + top + bottom, W + left + right) + img.shape[2:], img.dtype)
    out[...] = np.asarray(color, img.dtype) if img.ndim == 3 else color
    out[top:top + H, left:left + W] = img
    return out


def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab: [(dst index, src index, float32 weight)] in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(src, dsize):
    """src: float32 or uint8 [H, W] or [H, W, C]; dsize = (width, height) like cv2.resize."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]
    is_u8 = src.dtype == np.uint8
    s = src.astype(np.float32).reshape(sh, sw, -1)
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if scale_x >= 1 and scale_y >= 1:
        ix, iy = int(scale_x), int(scale_y)
        if abs(scale_x - ix) < np.finfo(np.float64).eps and abs(scale_y - iy) < np.finfo(np.float64).eps:
            # resizeAreaFast: row-major sum of the iy x ix block, times 1 / area
            acc = np.zeros((dh, dw, s.shape[2]), np.float32)
            for sy in range(iy):
                for sx in range(ix):
                    acc += s[sy:dh * iy:iy, sx:dw * ix:ix]
            out = acc * np.float32(1.0 / (ix * iy))
        else:
            xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
            out = np.zeros((dh, dw, s.shape[2]), np.float32)
            summ, prev = np.zeros((dw, s.shape[2]), np.float32), ytab[0][0]
            for (dy, sy, beta) in ytab:
                buf = np.zeros((dw, s.shape[2]), np.float32)
                for (dx, sx, alpha) in xtab:
                    buf[dx] += s[sy, sx] * alpha
                if dy != prev:
                    out[prev] = summ
                    summ, prev = beta * buf, dy
                else:
                    summ = summ + beta * buf
            out[prev] = summ
    else:
        # an enlarged axis: INTER_AREA runs the bilinear resizer with area-style coefficients (resize(): area_mode)
        def lin_tab(ssize, dsize, inv, scale):
            idx, w = np.zeros(dsize, np.int64), np.zeros(dsize, np.float32)
            for d in range(dsize):
                sx = math.floor(d * scale)
                fx = (d + 1) - (sx + 1) * inv
                fx = 0.0 if fx <= 0 else fx - math.floor(fx)
                if sx < 0:
                    sx, fx = 0, 0.0
                if sx >= ssize - 1:
                    sx, fx = ssize - 1, 0.0
                idx[d], w[d] = sx, np.float32(fx)
            return idx, w
        xi, xw = lin_tab(sw, dw, inv_x, scale_x)
        yi, yw = lin_tab(sh, dh, inv_y, scale_y)
        x1 = np.minimum(xi + 1, sw - 1)
        y1 = np.minimum(yi + 1, sh - 1)
        rows = s[:, xi] * (np.float32(1) - xw)[None, :, None] + s[:, x1] * xw[None, :, None]          # horizontal pass
        out = rows[yi] * (np.float32(1) - yw)[:, None, None] + rows[y1] * yw[:, None, None]           # vertical pass
    out = out.reshape((dh, dw) + src.shape[2:])
    if is_u8:
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out.astype(np.float32)
