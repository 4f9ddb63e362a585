"""A synthetic frame with the STRUCTURE of 8-bit photographs ("a k/255 quantised variant to exercise ties").

Every production input of the reference is uint8 (data.py:133-158, infer.py:35-40): values on the k/255 grid, spatially
coherent -- exact channel ties, exact zeros, saturated 255s and dark (< 0.04045) values over WHOLE wavefronts (256
consecutive pixels), none of which uniformly random floats have.  `coherent_8bit_frames` builds such frames, seeded and
size-independent, as uint8 [B,H,W,3]; the parity tests and `bench.py`'s `layer_8bit` row use the same generator.

Regions (bands of rows, each H/8 high; inside a band the content varies along the row):
  0  smooth colour gradients (three ramps of different slope)           -- the common case, no ties
  1  grey ramp r = g = b = k                                             -- three-way ties: hue terms all add (colors.py:221-224)
  2  flat dark patches: exact (0,0,0), (k,k,k) and (k,0,j) with k,j <= 10 -- below the sRGB threshold, zeros -> 1e-9 floor (colors.py:205)
  3  constant 64-pixel-wide patches from a palette with two-channel ties -- r=g>b, g=b>r, r=b>g, primaries, white, black
  4  low-frequency "photograph": a few sinusoids per channel + 1-LSB noise
  5  saturated highlights: one or two channels at 255, the third a ramp
  6  dark photograph: band 4 scaled to 0..40 (half of it below the threshold)
  7  vertical gradient x grey/colour checker (ties alternate with non-ties inside a wavefront)
"""
import numpy as np


def _band(kind, h, W, rng, phase):
    x = np.arange(W, dtype=np.float64)[None, :] / max(W - 1, 1)
    y = np.arange(h, dtype=np.float64)[:, None] / max(h - 1, 1)
    out = np.zeros((h, W, 3), np.float64)
    if kind == 0:
        out[..., 0] = 255 * x
        out[..., 1] = 255 * (1 - x) * (0.5 + 0.5 * y)
        out[..., 2] = 64 + 127 * y + 0 * x
    elif kind == 1:
        out[...] = (255 * np.abs(((x * 2 + phase) % 2) - 1))[..., None] + 0 * y[..., None]
    elif kind == 2:
        n = max(W // 96, 1)
        pal = np.array([[0, 0, 0], [3, 3, 3], [10, 10, 10], [7, 0, 2], [0, 9, 0], [1, 0, 10], [10, 10, 0], [0, 0, 0]], np.float64)
        idx = ((np.arange(W) // 96) + int(phase * 7)) % len(pal)
        out[...] = pal[idx][None, :, :]
        del n
    elif kind == 3:
        pal = np.array([[180, 180, 40], [40, 150, 150], [150, 40, 150], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255],
                        [0, 0, 0], [255, 255, 0], [0, 255, 255], [255, 0, 255], [128, 128, 128], [77, 77, 78], [200, 77, 77]], np.float64)
        idx = ((np.arange(W) // 64) + rng.randint(0, len(pal))) % len(pal)
        out[...] = pal[idx][None, :, :]
    elif kind in (4, 6):
        for c in range(3):
            acc = 0.5 + 0 * (x + y)
            for _ in range(4):
                fx, fy, ph, amp = rng.uniform(0.5, 6), rng.uniform(0.5, 4), rng.uniform(0, 6.28), rng.uniform(0.05, 0.2)
                acc = acc + amp * np.sin(2 * np.pi * (fx * x + fy * y) + ph)
            out[..., c] = 255 * np.clip(acc, 0, 1)
        out += rng.randint(-1, 2, size=out.shape)
        if kind == 6:
            out *= 40.0 / 255.0
    elif kind == 5:
        out[..., 0] = 255
        out[..., 1] = np.where(x < 0.5, 255, 255 * (1.5 - x)) + 0 * y
        out[..., 2] = 255 * x * y
    elif kind == 7:
        cell = ((np.arange(W)[None, :] // 2) + (np.arange(h)[:, None] // 2)) % 2
        v = 255 * y + 0 * x
        out[..., 0] = v
        out[..., 1] = np.where(cell == 0, v, 0.6 * v)
        out[..., 2] = np.where(cell == 0, v, 0.3 * v + 20)
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def coherent_8bit_frames(B, H, W, seed=0):
    """uint8 [B,H,W,3]; frame b rotates the band order by b so a batch does not repeat one image."""
    rng = np.random.RandomState(seed)
    frames = np.empty((B, H, W, 3), np.uint8)
    edges = [H * k // 8 for k in range(9)]
    for b in range(B):
        for k in range(8):
            r0, r1 = edges[k], edges[k + 1]
            if r1 > r0:
                frames[b, r0:r1] = _band((k + b) % 8, r1 - r0, W, rng, phase=0.37 * b)
    return frames


def describe(frames):
    """Fractions of pixels with the features random floats never have (for logs and DESIGN.md)."""
    a = frames.reshape(-1, 3).astype(np.int32)
    ties = ((a[:, 0] == a[:, 1]) | (a[:, 1] == a[:, 2]) | (a[:, 0] == a[:, 2])).mean()
    return {"ties": float(ties), "zero_channel": float((a == 0).any(1).mean()), "dark_channel": float((a <= 10).any(1).mean()),
            "saturated_channel": float((a == 255).any(1).mean())}


if __name__ == "__main__":
    f = coherent_8bit_frames(2, 1000, 1500)
    print(f.shape, describe(f))
