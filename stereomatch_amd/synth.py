"""Deterministic synthetic rectified stereo pairs: identical bytes on every
machine, no library RNG.  splitmix64 in counter mode."""
import numpy as np

SEED_SCENE = 0x5EED0001
SEED_NOISE = 0x5EED0002
_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    """n successive outputs of splitmix64 seeded with `seed`, skipping `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _value_noise(w, h, cell, seed, offset):
    gw, gh = w // cell + 2, h // cell + 2
    g = (splitmix64(seed, gw * gh, offset) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    g = g.reshape(gh, gw)
    ys, xs = np.arange(h) / cell, np.arange(w) / cell
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy, gw * gh


def scene_pair(w, h, n_disp, n_bands=1, seed=SEED_SCENE):
    """Left (reference) / right (matching) uint8 images [h, w, n_bands] and the ground-truth
    disparity of the right view: 4 octaves of bilinear value noise; slanted background plus
    6 fronto-parallel rectangles; right = left warped by the disparity + uniform noise +-2."""
    off = 0
    acc = np.zeros((h, w))
    amp, total = 1.0, 0.0
    for cell in (64, 16, 4, 1):
        n, used = _value_noise(w, h, cell, seed, off)
        off += used
        acc += amp * n
        total += amp
        amp *= 0.5
    left = np.clip(acc / total * 255.0 + 0.5, 0, 255).astype(np.uint8)

    dmax = n_disp - 1
    xs = np.arange(w)[None, :]
    disp = np.broadcast_to((dmax * 0.15 + dmax * 0.25 * xs / max(w - 1, 1)), (h, w)).copy()
    r = splitmix64(seed, 6 * 5, off)
    off += 30
    for i in range(6):
        q = (r[5 * i:5 * i + 5] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        x0, y0 = int(q[0] * w * 0.8), int(q[1] * h * 0.8)
        rw, rh = int(w * (0.08 + 0.2 * q[2])), int(h * (0.08 + 0.2 * q[3]))
        disp[y0:y0 + rh, x0:x0 + rw] = 0.3 * dmax + 0.7 * dmax * q[4]
    disp = np.clip(np.rint(disp), 0, dmax).astype(np.int64)

    src = np.clip(xs + disp, 0, w - 1)
    right = np.take_along_axis(left, src, axis=1).astype(np.int64)
    noise = (splitmix64(seed, w * h, off) % np.uint64(5)).astype(np.int64).reshape(h, w) - 2
    right = np.clip(right + noise, 0, 255).astype(np.uint8)
    return _bands(left, n_bands), _bands(right, n_bands), disp.astype(np.int32)


def noise_pair(w, h, n_bands=1, seed=SEED_NOISE):
    """Two independent uniform-random uint8 images (the no-structure case)."""
    n = w * h * n_bands
    v = (splitmix64(seed, 2 * n) >> np.uint64(56)).astype(np.uint8)
    return v[:n].reshape(h, w, n_bands), v[n:].reshape(h, w, n_bands)


def _bands(img, n_bands):
    if n_bands == 1:
        return img[:, :, None].copy()
    out = np.empty(img.shape + (4,), np.uint8)
    # B, G, R from the gray value with fixed per-band offsets; A = 255 (ImageIO.cpp:390-396)
    out[..., 0] = img
    out[..., 1] = np.clip(img.astype(np.int64) + 3, 0, 255).astype(np.uint8)
    out[..., 2] = 255 - img
    out[..., 3] = 255
    return out
