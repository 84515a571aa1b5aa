"""Seeded synthetic inputs.  No datasets exist offline, so
every test/bench input is regenerated from an integer seed with numpy PCG64."""
import numpy as np


def make_image(seed: int, w: int = 640, h: int = 480) -> np.ndarray:
    """C1 generator: mid-gray + 400 random rectangles + 200 filled discs + +-4 noise
    (counts scale with image area so 1280x720 frames keep the same corner density)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    area = (w * h) / (640.0 * 480.0)
    img = np.full((h, w), 128, dtype=np.int32)
    n_rect, n_disc = int(round(400 * area)), int(round(200 * area))
    for _ in range(n_rect):
        rw, rh = rng.integers(8, 65, size=2)
        x0 = int(rng.integers(-16, w))
        y0 = int(rng.integers(-16, h))
        g = int(rng.integers(0, 256))
        img[max(y0, 0):max(y0 + int(rh), 0), max(x0, 0):max(x0 + int(rw), 0)] = g
    for _ in range(n_disc):
        r = int(rng.integers(3, 13))
        cx = int(rng.integers(0, w))
        cy = int(rng.integers(0, h))
        g = int(rng.integers(0, 256))
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = g
    img += rng.integers(-4, 5, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def make_batch(seed0: int, n: int, w: int = 640, h: int = 480) -> np.ndarray:
    return np.stack([make_image(seed0 + i, w, h) for i in range(n)])
