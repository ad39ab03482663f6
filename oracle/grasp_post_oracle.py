"""CPU restatement (numpy) of the reference planner's grasp post-processing -- TEST INFRASTRUCTURE ONLY (see
oracle/graspnerf_oracle.py for the rules: imported by tests/, never by the product).

The reference (src/nr/main.py:23-84) calls scipy.ndimage (requirements.txt:13, unpinned; 1.15.3 in this image), a
third-party dependency outside /root/reference.  The three filters are restated here from scipy's published algorithms:
  * gaussian_filter(sigma, mode='nearest'): separable; per axis a correlation with w[k] = exp(-k^2/(2 sigma^2)) / sum,
    radius int(4 sigma + .5), accumulated in float64 as  centre term, then (in[-k] + in[+k]) * w[k] for k = r..1
    (ni_filters.c, symmetric branch), result rounded to the array dtype (float32) after EVERY axis;
  * binary_dilation(structure = 6-neighbourhood, iterations, mask, border_value=0): x <- where(mask, dilate(x), x);
  * maximum_filter(size=4, mode='reflect'): window [i-2, i+1] per axis, borders mirrored about the edge sample.
Pinned by tests/golden/golden_post.npz, produced by the reference's own process()/select() (tools/make_goldens.py
--post-only)."""
import numpy as np


def gaussian_kernel1d(sigma, truncate=4.0):
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return w / w.sum(), r


def gaussian_filter_nearest(vol, sigma):
    w, r = gaussian_kernel1d(sigma)
    out = np.asarray(vol, np.float32).copy()
    for ax in range(out.ndim):
        n = out.shape[ax]
        src = np.moveaxis(out, ax, -1).astype(np.float64)
        idx = np.arange(n)
        acc = src * w[r]
        for k in range(r, 0, -1):
            lo = np.take(src, np.clip(idx - k, 0, n - 1), axis=-1)
            hi = np.take(src, np.clip(idx + k, 0, n - 1), axis=-1)
            acc = acc + (lo + hi) * w[r - k]
        out = np.moveaxis(acc.astype(np.float32), -1, ax).copy()
    return out


def dilate6(x):
    p = np.pad(x, 1, constant_values=False)
    c = p[1:-1, 1:-1, 1:-1]
    return (c | p[:-2, 1:-1, 1:-1] | p[2:, 1:-1, 1:-1] | p[1:-1, :-2, 1:-1] | p[1:-1, 2:, 1:-1]
            | p[1:-1, 1:-1, :-2] | p[1:-1, 1:-1, 2:])


def masked_dilation(x, mask, iterations):
    x = x.copy()
    for _ in range(iterations):
        x = np.where(mask, dilate6(x), x)
    return x


def maximum_filter_reflect(vol, size=4):
    out = vol
    left, right = size // 2, size - size // 2 - 1
    for ax in range(vol.ndim):
        n = out.shape[ax]
        idx = np.arange(n)
        best = None
        for d in range(-left, right + 1):
            j = idx + d
            j = np.where(j < 0, -j - 1, j)
            j = np.where(j >= n, 2 * n - 1 - j, j)
            v = np.take(out, j, axis=ax)
            best = v if best is None else np.maximum(best, v)
        out = best
    return out


def process(tsdf, qual, rot, width, sigma=1.0, min_width=1.33, max_width=9.33, thres_high=0.5, thres_low=1e-3):
    """ref: main.py:23-57.  [R,R,R] float32 volumes (rot [4,R,R,R]) -> processed quality volume."""
    f = np.float32
    q = gaussian_filter_nearest(qual, sigma)
    outside = tsdf > f(thres_high)
    inside = (f(thres_low) < tsdf) & (tsdf < f(thres_high))
    valid = masked_dilation(outside, ~inside, 2)
    q[~valid] = 0.0
    q[(width < f(min_width)) | (width > f(max_width))] = 0.0
    return q


def select(qual, rot, width, threshold=0.90, size=4):
    """ref: main.py:60-84.  -> index [N,3] (argwhere order), score [N], quat [N,4], width [N]."""
    q = qual.copy()
    q[q < np.float32(threshold)] = 0.0
    m = maximum_filter_reflect(q, size)
    q = np.where(q == m, q, np.float32(0.0))
    idx = np.argwhere(q != 0)
    i, j, k = idx[:, 0], idx[:, 1], idx[:, 2]
    return idx, q[i, j, k], rot[:, i, j, k].T, width[i, j, k]
