"""Integer-only synthetic I420 clips: smooth translating
texture + a moving 64x64 box + +-3 LSB hash noise.  uint32 wrap-around
arithmetic so the same frames can be regenerated anywhere (C, numpy)."""
import numpy as np

SEED = 1234


def _tri(v):
    a = v & 255
    return np.where(a < 128, a.astype(np.int64) - 64, 191 - a.astype(np.int64))


def _noise(x, y, t, p, amp):
    with np.errstate(over="ignore"):
        h = (x.astype(np.uint32) * np.uint32(73856093)) ^ (y.astype(np.uint32) * np.uint32(19349663)) \
            ^ np.uint32((t * 83492791) & 0xFFFFFFFF) ^ np.uint32((p * 2654435761) & 0xFFFFFFFF) ^ np.uint32(SEED)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0x5BD1E995)
        h ^= h >> np.uint32(15)
    return (h % np.uint32(2 * amp + 1)).astype(np.int64) - amp


def frame(w, h, t):
    """Return (Y[h][w], U[h/2][w/2], V[h/2][w/2]) uint8 for frame index t."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    lum = 128 + (_tri((x + 3 * t) * 4) >> 1) + (_tri((y - 2 * t) * 6) >> 2) + (_tri((x + y + 5 * t) * 9) >> 3)
    bx, by = (40 + 7 * t) % (w - 64), (30 + 3 * t) % (h - 64)
    inside = (x >= bx) & (x < bx + 64) & (y >= by) & (y < by + 64)
    lum = lum + np.where(inside, _tri(x * 16) >> 1, 0) + _noise(x, y, t, 0, 3)
    yc, xc = np.mgrid[0:h // 2, 0:w // 2].astype(np.int64)
    u = 128 + (_tri((xc + 2 * t) * 3) >> 2) + _noise(xc, yc, t, 1, 1)
    v = 128 + (_tri((yc - t) * 5) >> 2) + _noise(xc, yc, t, 2, 1)
    c = lambda a: np.clip(a, 0, 255).astype(np.uint8)
    return c(lum), c(u), c(v)
