"""Deterministic synthetic frame sources.

The reference's only frame source is a webcam (src/io/webcam.cpp), which a headless MI355X node
does not have; these generators stand in for it in the tests and in bench.py.  Everything is
integer arithmetic on a splitmix64 stream so any other language can reproduce the frames bit for
bit: seed = 0xC0FFEE + frame_index.
"""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = 0x9E3779B97F4A7C15
SEED0 = 0xC0FFEE


def splitmix64(seed, n):
    """n 64-bit outputs of splitmix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _bytes(seed, n):
    words = splitmix64(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n]


def noise(w, h, seed=SEED0):
    """iid uniform u8."""
    return _bytes(seed, w * h).reshape(h, w).copy()


def flat(w, h, v):
    return np.full((h, w), v, np.uint8)


def steps(w, h, height, kind="vertical", base=0):
    """A single step edge of the given height through the middle of the frame."""
    img = np.full((h, w), base, np.int32)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "vertical":
        img[xx >= w // 2] += height
    elif kind == "horizontal":
        img[yy >= h // 2] += height
    else:  # diagonal
        img[(xx - w // 2) + (yy - h // 2) >= 0] += height
    return np.clip(img, 0, 255).astype(np.uint8)


def natural(w, h, seed=SEED0, nshapes=40, sigma_q=7):
    """Linear-gradient background + random filled rectangles/ellipses + approx. Gaussian noise.

    noise = ((b0+b1+b2+b3 - 510) * sigma_q) >> 8 with b_i uniform bytes: sigma ~= 0.577*sigma_q (4.04
    for the default 7).  Edge density after Mode R Canny at 10/40 is a few percent.
    """
    r = splitmix64(seed ^ 0x5851F42D4C957F2D, 8 * nshapes + 8).astype(np.uint64)
    yy, xx = np.mgrid[0:h, 0:w]
    g0 = int(r[0] % np.uint64(96)) + 32
    gx = int(r[1] % np.uint64(97)) - 48
    gy = int(r[2] % np.uint64(97)) - 48
    img = g0 + (gx * xx) // max(w, 1) + (gy * yy) // max(h, 1)
    img = img.astype(np.int32)
    for s in range(nshapes):
        q = r[8 + 8 * s: 16 + 8 * s]
        cx = int(q[0] % np.uint64(w))
        cy = int(q[1] % np.uint64(h))
        rw = int(q[2] % np.uint64(max(w // 4, 2))) + 2
        rh = int(q[3] % np.uint64(max(h // 4, 2))) + 2
        level = int(q[4] % np.uint64(256))
        x0, x1 = max(cx - rw, 0), min(cx + rw, w)
        y0, y1 = max(cy - rh, 0), min(cy + rh, h)
        if x0 >= x1 or y0 >= y1:
            continue
        if int(q[5] & np.uint64(1)):
            img[y0:y1, x0:x1] = level
        else:
            sub_y, sub_x = np.mgrid[y0:y1, x0:x1]
            m = ((sub_x - cx) * (sub_x - cx)) * (rh * rh) + ((sub_y - cy) * (sub_y - cy)) * (rw * rw) <= (rw * rw) * (rh * rh)
            img[y0:y1, x0:x1][m] = level
    b = _bytes(seed ^ 0x2545F4914F6CDD1D, 4 * w * h).reshape(4, h, w).astype(np.int32)
    nz = ((b.sum(axis=0) - 510) * sigma_q) >> 8
    return np.clip(img + nz, 0, 255).astype(np.uint8)


def serpentine(w, h, amp=20, seed_amp=120, band=6, pitch=24, margin=12):
    """Adversarial hysteresis input: one long boustrophedon band whose blurred gradient stays
    between the default thresholds (candidate only), with a short high-contrast head (strong
    seed) at one end.  Every edge pixel of the band must be reached from that one seed."""
    img = np.zeros((h, w), np.uint8)
    y = margin
    left = True
    first = True
    while y + band + pitch < h - margin:
        img[y:y + band, margin:w - margin] = amp
        if first:
            img[y:y + band, margin:margin + 16] = seed_amp
            first = False
        x0 = (w - margin - band) if left else margin
        img[y:y + pitch + band, x0:x0 + band] = amp
        left = not left
        y += pitch
    img[y:y + band, margin:w - margin] = amp
    return img


def thresh_map_serpentine(w, h):
    """Tri-state (0/128/255) map: a 1-px candidate boustrophedon path (horizontal runs on odd rows,
    single-pixel connectors at alternating ends) with one strong pixel at its start -- the worst
    case for tile-iterated hysteresis: the whole path must be reached from that one seed."""
    t = np.zeros((h, w), np.uint8)
    left = True
    for y in range(1, h - 1, 2):
        t[y, 1:w - 1] = 128
        if y + 2 < h - 1:
            t[y + 1, (w - 2) if left else 1] = 128
        left = not left
    t[1, 1] = 255
    return t


def thresh_map_random(w, h, seed=SEED0, p_cand=0.30, p_strong=0.01):
    """iid tri-state map: dense candidate clutter with sparse strong seeds."""
    b = _bytes(seed ^ 0x1234567, w * h).reshape(h, w).astype(np.int32)
    t = np.zeros((h, w), np.uint8)
    t[b < int(256 * (p_cand + p_strong))] = 128
    t[b < max(int(256 * p_strong), 1)] = 255
    return t


def frames(kind, w, h, n, seed=SEED0):
    """n frames (n,h,w) u8 of a generator; frame i uses seed + i."""
    out = np.empty((n, h, w), np.uint8)
    for i in range(n):
        if kind == "noise":
            out[i] = noise(w, h, seed + i)
        elif kind == "natural":
            out[i] = natural(w, h, seed + i)
        else:
            raise ValueError(kind)
    return out
