"""ORACLE — TEST INFRASTRUCTURE ONLY.  A stand-in for the handful of `cv2` calls made by the reference functions that
tests execute from /root/reference/tools/util.py via `ast` (cv2 itself is not in this image): written on scipy.ndimage /
plain NumPy, independently of mimo_amd/cvops.py, from the documented OpenCV semantics of each call.

  cvtColor(img, COLOR_RGB2GRAY)          8-bit fixed point, OpenCV 4.x: (9798 R + 19235 G + 3735 B + 16384) >> 15
  getStructuringElement(MORPH_RECT, k)   all-ones k x k
  morphologyEx(m, MORPH_CLOSE | MORPH_OPEN, se)   window anchored at k // 2, border never contributes
  boundingRect(m)                        (x, y, w, h) of the non-zero pixels, (0, 0, 0, 0) if none
  copyMakeBorder(..., BORDER_CONSTANT, value)
"""
import numpy as np
from scipy import ndimage

COLOR_RGB2GRAY, MORPH_RECT, MORPH_CLOSE, MORPH_OPEN, BORDER_CONSTANT = 7, 0, 3, 2, 0


def cvtColor(img, code):
    assert code == COLOR_RGB2GRAY and img.dtype == np.uint8
    out = np.empty(img.shape[:2], np.uint8)
    flat = img.reshape(-1, 3).tolist()
    out.reshape(-1)[:] = [(9798 * r + 19235 * g + 3735 * b + 16384) // 32768 for r, g, b in flat]
    return out


def getStructuringElement(shape, ksize):
    assert shape == MORPH_RECT
    return np.ones((ksize[1], ksize[0]), np.uint8)


def _dil(m, k):
    return ndimage.maximum_filter(m, size=k, mode="constant", cval=0)


def _ero(m, k):
    return ndimage.minimum_filter(m, size=k, mode="constant", cval=255)


def morphologyEx(m, op, se):
    k = se.shape
    if op == MORPH_CLOSE:
        return _ero(_dil(m, k), k)
    if op == MORPH_OPEN:
        return _dil(_ero(m, k), k)
    raise NotImplementedError(op)


def boundingRect(m):
    rows, cols = np.flatnonzero(m.any(axis=1)), np.flatnonzero(m.any(axis=0))
    if rows.size == 0:
        return 0, 0, 0, 0
    return int(cols[0]), int(rows[0]), int(cols[-1] - cols[0] + 1), int(rows[-1] - rows[0] + 1)


def copyMakeBorder(img, top, bottom, left, right, borderType, value=None):
    assert borderType == BORDER_CONSTANT
    out = np.pad(img, ((top, bottom), (left, right)) + ((0, 0),) * (img.ndim - 2))
    v = np.asarray(value, img.dtype)
    out[:top] = v
    out[out.shape[0] - bottom:] = v
    out[:, :left] = v
    out[:, out.shape[1] - right:] = v
    return out
