"""Seeded synthetic inputs: checkpoints in the reference's layout and frames.

The trained snapshots the reference names (snapshots/super_point.pt,
snapshots/magic_point.pt) are not shipped with it, and its frame generator
(python/src/synthetic_shapes.py) needs OpenCV.  Everything the tests and the
bench feed the path is therefore generated here, from numpy's PCG64 only, so the
same seed gives the same bytes in this container and on the GPU box.

`make_state_dict` produces the 163 entries of ``ckpt['model_state_dict']``
(arch.state_dict_spec) with He-scaled conv weights and *non-trivial* BatchNorm
statistics (a folding bug is invisible with mean 0 / var 1 / gamma 1 / beta 0),
plus the "dustbin" bias knob of SURVEY.md section 8(d) that sets how many pixels
pass the 0.015 confidence threshold.
"""
import numpy as np

from . import arch


def make_state_dict(seed=0, dustbin_bias=2.0, gain=1.0):
    """{name: np.ndarray} in checkpoint order; float32 (num_batches_tracked int64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in arch.state_dict_spec().items():
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(1000, dtype=np.int64)
        elif name.endswith("up_sample.bias"):
            sd[name] = rng.normal(0.0, 0.05, shape).astype(np.float32)
        elif len(shape) == 4:
            if name.endswith("up_sample.weight"):
                # each output pixel of the stride-2 transposed conv sees 9/4 taps on average
                fan_in = shape[0] * 9 / 4.0
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            b = gain * np.sqrt(6.0 / fan_in)
            # the 1x1 conv that closes a block and the projection shortcut are summed:
            # halve their variance so the sum keeps the scale of its inputs
            if ".conv2." in name or "identity_downsample" in name:
                b *= np.sqrt(0.5)
            sd[name] = rng.uniform(-b, b, shape).astype(np.float32)
        elif leaf == "weight":
            sd[name] = rng.uniform(0.75, 1.25, shape).astype(np.float32)
        elif leaf == "bias":
            sd[name] = rng.normal(0.0, 0.1, shape).astype(np.float32)
        elif leaf == "running_mean":
            sd[name] = rng.normal(0.0, 0.1, shape).astype(np.float32)
        elif leaf == "running_var":
            sd[name] = rng.uniform(0.75, 1.25, shape).astype(np.float32)
        else:
            raise AssertionError(name)
    sd["detector.layer.1.bn2.bias"][64] += np.float32(dustbin_bias)
    return sd


def make_vgg_state_dict(seed=0, dustbin_bias=3.0):
    """Seeded parameters of the reference's C++ network (superpoint::SPModel, cpp/src/model.cc) as the flat
    {name: float32 array} dict cpp/src/superpoint.cc:27-55 loads: He-uniform weights (every layer keeps the
    activation scale), small biases, and the dustbin knob on detector_conv_b.bias[64]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in arch.vgg_state_dict_spec().items():
        if len(shape) == 4:
            b = np.sqrt(6.0 / (shape[1] * shape[2] * shape[3]))
            sd[name] = rng.uniform(-b, b, shape).astype(np.float32)
        else:
            sd[name] = rng.normal(0.0, 0.05, shape).astype(np.float32)
    sd["detector_conv_b.bias"][64] += np.float32(dustbin_bias)
    return sd


def make_frame(seed, h=480, w=640, gray=False):
    """HxWx3 float32 image in [0,1]: 40 random filled rectangles / triangles on a
    flat background, 5x5 box-blurred.
    `gray=True` gives a gray image replicated over the 3 channels, which is what
    the reference does with gray inputs (python/src/dataset_utils.py:19-20)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nc = 1 if gray else 3
    img = np.empty((h, w, nc), dtype=np.float32)
    img[:] = rng.uniform(0.2, 0.8, nc).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(40):
        col = rng.uniform(0.0, 1.0, nc).astype(np.float32)
        if rng.integers(0, 2) == 0:
            x0, x1 = np.sort(rng.integers(0, w, 2))
            y0, y1 = np.sort(rng.integers(0, h, 2))
            img[y0:y1 + 1, x0:x1 + 1] = col
        else:
            cx, cy = rng.integers(0, w), rng.integers(0, h)
            r = max(h, w) // 4
            px = cx + rng.integers(-r, r + 1, 3)
            py = cy + rng.integers(-r, r + 1, 3)
            lo_y, hi_y = max(int(py.min()), 0), min(int(py.max()), h - 1)
            lo_x, hi_x = max(int(px.min()), 0), min(int(px.max()), w - 1)
            if hi_y < lo_y or hi_x < lo_x:
                continue
            sy, sx = yy[lo_y:hi_y + 1, lo_x:hi_x + 1], xx[lo_y:hi_y + 1, lo_x:hi_x + 1]

            def edge(i, j):
                return (px[j] - px[i]) * (sy - py[i]) - (py[j] - py[i]) * (sx - px[i])
            e0, e1, e2 = edge(0, 1), edge(1, 2), edge(2, 0)
            inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
            img[lo_y:hi_y + 1, lo_x:hi_x + 1][inside] = col
    # 5x5 box blur with edge replication, as two separable passes in float64
    a = img.astype(np.float64)
    p = np.pad(a, ((2, 2), (0, 0), (0, 0)), mode="edge")
    a = sum(p[i:i + h] for i in range(5)) / 5.0
    p = np.pad(a, ((0, 0), (2, 2), (0, 0)), mode="edge")
    a = sum(p[:, i:i + w] for i in range(5)) / 5.0
    out = np.clip(a, 0.0, 1.0).astype(np.float32)
    if gray:
        out = np.repeat(out, 3, axis=2)
    return out


def make_batch(first_seed, n, h=480, w=640, gray=False):
    """float32 [n,3,h,w] (NCHW, the layout SuperPoint.forward takes:
    python/src/superpoint.py:91, python/src/inferencewrapper.py:70-81)."""
    out = np.empty((n, 3, h, w), dtype=np.float32)
    for i in range(n):
        out[i] = make_frame(first_seed + i, h, w, gray).transpose(2, 0, 1)
    return out


def scale_activations(sd, s):
    """A checkpoint whose EVERY activation is `s` times that of `sd` on the same frames (exactly, in exact arithmetic):
    the stem's BatchNorm gets weight and bias times s, every later BatchNorm running_mean and bias times s (bn(s y) with
    mean s mu and bias s beta is s bn(y); ReLU, max-pool, the shortcut sums and the concat are positively homogeneous),
    and so does the transposed convolution's bias.  Logits and descriptor map come out times s: what a trained network
    with a larger dynamic range than the He-scaled synthetic ones looks like to the arithmetic."""
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    f = np.float32(s)
    for name in out:
        leaf = name.rsplit(".", 1)[1]
        if name.startswith("encoder.bn1."):
            if leaf in ("weight", "bias"):
                out[name] = (out[name] * f).astype(np.float32)
        elif leaf in ("running_mean",) or (leaf == "bias" and out[name].ndim == 1):
            out[name] = (out[name] * f).astype(np.float32)
    return out
