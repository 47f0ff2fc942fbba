"""Host-side mirror of the reference's Python interface for this path:

    reference                                      here
    python/src/settings.py  SuperPointSettings  -> SuperPointSettings
    python/src/saveutils.py load_checkpoint_for_inference -> load_checkpoint_for_inference
    python/src/superpoint.py SuperPoint.forward -> SuperPoint.forward / __call__
    python/src/netutils.py  get_points          -> get_points
    python/src/netutils.py  get_descriptors     -> get_descriptors
    python/src/inferencewrapper.py InferenceWrapper.{run,prepare_input} -> same names

Same names, argument meaning and return conventions (points: float64 [3,K] rows
x, y, confidence, descending confidence; descriptors: float32 [D,K], unit columns).
Differences, all documented in DESIGN.md: load failures raise instead of exit();
batches give independent per-frame results (the reference merges frames,
netutils.py:59-61); ties in confidence are ordered by row-major pixel index.
"""
import os

import numpy as np
import torch

from .engine import Engine


class SuperPointSettings:
    """Inference fields of python/src/settings.py:2-8 (training fields are out of scope)."""

    def __init__(self):
        self.cuda = True
        self.nms_dist = 4
        self.confidence_thresh = 0.015
        self.nn_thresh = 0.7
        self.cell = 8
        self.border_remove = 4
        # not in the reference: the arithmetic mode of the kernels (include/fpc.h FPC_F32 / FPC_F32_SPLIT / ...)
        self.dtype = "f32"


def load_checkpoint_for_inference(filename):
    """Reads the reference's checkpoint file (saveutils.py:54-63: a dict with
    'model_state_dict') and returns the state dict; a bare state dict is accepted too."""
    if not os.path.exists(filename):
        raise FileNotFoundError("Failed to load checkpoint: %s" % filename)
    ckpt = torch.load(filename, map_location="cpu", weights_only=True)
    if isinstance(ckpt, dict) and "model_state_dict" in ckpt:
        ckpt = ckpt["model_state_dict"]
    return ckpt


class SuperPoint:
    """The network object (python/src/superpoint.py:64-115) backed by libfpc.so."""

    def __init__(self, settings, device=0, max_batch=1):
        self.settings = settings
        self.is_descriptor_enabled = True
        self.device = device
        self.max_batch = max_batch
        self._state_dict = None
        self._engines = {}

    def disable_descriptor(self):       # superpoint.py:74-78
        self.is_descriptor_enabled = False

    def enable_descriptor(self):        # superpoint.py:80-84
        self.is_descriptor_enabled = True

    def load_state_dict(self, state_dict, strict=True):
        self._state_dict = state_dict
        for e in self._engines.values():
            e.load_state_dict(state_dict)

    def engine(self, h, w, batch=1):
        key = (h, w, self.is_descriptor_enabled)
        e = self._engines.get(key)
        if e is None or e.max_batch < batch:
            if e is not None:
                e.close()
            s = self.settings
            if self._state_dict is None:
                raise RuntimeError("SuperPoint: no weights loaded")
            # the flat dict of the reference's C++ network (cpp/src/model.cc) selects that architecture (gray frames)
            vgg = "encoder_conv0_a.weight" in self._state_dict
            e = Engine(h, w, max(batch, self.max_batch), self.device, s.nms_dist, s.confidence_thresh,
                       s.border_remove, self.is_descriptor_enabled, in_channels=1 if vgg else 3,
                       dtype=getattr(s, "dtype", "f32"), arch="vgg" if vgg else "resnet")
            e.load_state_dict(self._state_dict)
            self._engines[key] = e
        return e

    def forward(self, image):
        if len(image.shape) <= 2:       # superpoint.py:94-95
            return torch.empty((1,)), torch.empty((1,)), torch.empty((1,))
        h, w = image.shape[-2:]
        return self.engine(h, w, image.shape[0]).forward(image)

    __call__ = forward


class HomographyConfig:
    """python/src/homographies.py:33-61 (same fields and defaults)."""

    def __init__(self):
        self.num = 15
        self.perspective = True
        self.scaling = True
        self.rotation = True
        self.translation = True
        self.n_scales = 5
        self.n_angles = 25
        self.scaling_amplitude = 0.1
        self.perspective_amplitude_x = 0.1
        self.perspective_amplitude_y = 0.1
        self.patch_ratio = 0.5
        self.max_angle = np.pi / 2
        self.allow_artifacts = False
        self.translation_overflow = 0.
        self.valid_border_margin = 8
        self.aggregation = "sum"

    def init_for_preprocess(self):
        self.translation = self.rotation = self.scaling = self.perspective = True
        self.scaling_amplitude = 0.2
        self.perspective_amplitude_x = 0.2
        self.perspective_amplitude_y = 0.2
        self.allow_artifacts = True
        self.patch_ratio = 0.85


def truncated_normal(n, mean=0.0, stddev=1.0, rng=None):
    """truncated_normal (python/src/homographies.py:64-67) AS THE REFERENCE REALLY DRAWS IT: scipy's
    `truncnorm(a, b)` with a = mean - 2 sigma, b = mean + 2 sigma and NO loc / scale -- i.e. a STANDARD normal cut to
    the interval [mean - 2 sigma, mean + 2 sigma], not N(mean, sigma^2) cut at +-2 sigma.  For the sigmas its caller
    passes (0.05 ... 0.1) that is nearly uniform over the interval (for the scales, centred on 1: a slightly falling
    density on [0.8, 1.2]).  Reproduced, not corrected: the label distribution of homography adaptation depends on it
    (round 2 drew the textbook distribution instead; SURVEY appendix B has no line for this quirk).  float32 [n]."""
    from scipy.stats import truncnorm
    if not stddev > 0:
        return np.full(n, mean, np.float32)
    return truncnorm(mean - 2 * stddev, mean + 2 * stddev).rvs(n, random_state=rng).astype(np.float32)


def sample_homography(shape, config=None, rng=None, reference_aliasing=True):
    """Restatement of `homographies.py:78-192` (python/src), quirks included -- a numpy transliteration in the reference's
    own step order, not an independent design: a centred patch is perturbed (perspective, scale, translation,
    rotation), each step keeping the corners inside the unit square unless allow_artifacts, and the 8 coefficients
    mapping output points to input points are solved for.

    REPRODUCED QUIRK (round 4; the review measured it on the reference's own function): `pts2 = pts1` (:117) is an
    ALIAS, and the perspective step (:127) and the translation step (:155) perturb it IN PLACE, while the scaling
    (:142) and rotation (:174) steps rebind `pts2` to a new tensor.  So the output corners `pts1` receive the
    perspective displacement too (and the translation, when scaling is off), and the transform solved from
    pts1 -> pts2 is a SIMILARITY (scale, rotation, translation): h7 = h8 = 0 up to rounding for every configuration
    -- no perspective term is ever drawn, whatever `perspective_amplitude_*` says.  The same aliasing makes `:179-180`
    multiply the shared corners by the frame size twice when neither scaling nor rotation rebinds pts2 (the solve is
    then the identity).  numpy's in-place operators and basic-index views alias exactly like torch's here, so the
    statements below are written with the reference's own mix of `+=` and rebinding.  `reference_aliasing=False` gives
    the sampler the docstring of the reference describes (pts1 stays the centred patch: real perspective terms).

    Random numbers come from `rng` (numpy Generator) instead of torch's / scipy's global state, so a run is reproducible
    from a seed but not bit-identical to the reference's stream (fixture F10 holds the reference's draws; the CPU test
    compares distributions).  shape = (H, W).  -> float32 [8]."""
    cfg = config or HomographyConfig()
    rng = rng or np.random.default_rng()

    def tn(n, mean, std):       # truncated_normal :64-67 (see above: a standard normal cut to mean +- 2 sigma)
        return truncated_normal(n, mean, std, rng)

    margin = (1 - cfg.patch_ratio) / 2
    pts1 = margin + np.array([[0, 0], [0, cfg.patch_ratio], [cfg.patch_ratio, cfg.patch_ratio], [cfg.patch_ratio, 0]], np.float32)
    pts2 = pts1 if reference_aliasing else pts1.copy()          # :117 `pts2 = pts1`
    if cfg.perspective:
        ax_, ay_ = cfg.perspective_amplitude_x, cfg.perspective_amplitude_y
        if not cfg.allow_artifacts:
            ax_, ay_ = min(ax_, margin), min(ay_, margin)
        pd, hl, hr = tn(1, 0., ay_ / 2)[0], tn(1, 0., ax_ / 2)[0], tn(1, 0., ax_ / 2)[0]
        pts2 += np.array([[hl, pd], [hl, -pd], [hr, pd], [hr, -pd]], np.float32)      # :127 in place: pts1 moves too
    if cfg.scaling:
        scales = np.concatenate([[1.], tn(cfg.n_scales, 1, cfg.scaling_amplitude / 2)]).astype(np.float32)
        center = pts2.mean(0, keepdims=True)
        scaled = (pts2 - center)[None] * scales[:, None, None] + center
        valid = np.arange(cfg.n_scales) if cfg.allow_artifacts else np.nonzero(((scaled >= 0.) & (scaled < 1.)).sum((1, 2)))[0]
        pts2 = scaled[valid[rng.integers(len(valid))]]          # :142 rebinds: from here on pts2 is its own array
    if cfg.translation:
        t_min, t_max = pts2.min(0), (1. - pts2).min(0)
        if cfg.allow_artifacts:
            t_min, t_max = t_min + cfg.translation_overflow, t_max + cfg.translation_overflow

        def uni(lo, hi):        # random_uniform :70-75
            lo, hi = (hi, lo) if lo > hi else (lo, hi)
            return rng.uniform(lo, hi if hi > lo else lo + 0.00001)
        pts2 += np.array([[uni(-t_min[0], t_max[0]), uni(-t_min[1], t_max[1])]], np.float32)      # :155 in place
    if cfg.rotation:
        angles = np.concatenate([[0.], np.linspace(-cfg.max_angle, cfg.max_angle, cfg.n_angles)]).astype(np.float32)
        center = pts2.mean(0, keepdims=True)
        rot = np.stack([np.cos(angles), -np.sin(angles), np.sin(angles), np.cos(angles)], 1).reshape(-1, 2, 2)
        rotated = np.matmul(np.tile((pts2 - center)[None], (cfg.n_angles + 1, 1, 1)), rot) + center
        valid = np.arange(cfg.n_angles) if cfg.allow_artifacts else np.nonzero(((rotated >= 0.) & (rotated < 1.)).sum((1, 2)))[0]
        pts2 = rotated[valid[rng.integers(len(valid))]]         # :174 rebinds
    size = np.array(shape[::-1], np.float32)[None]      # (W, H)
    pts1 *= size                                        # :179-180, in place as there (twice on a still-shared array)
    pts2 *= size
    p1, p2 = pts1, pts2
    a_mat = np.array([f(p1[i], p2[i]) for i in range(4) for f in (
        lambda p, q: [p[0], p[1], 1, 0, 0, 0, -p[0] * q[0], -p[1] * q[0]],
        lambda p, q: [0, 0, 0, p[0], p[1], 1, -p[0] * q[1], -p[1] * q[1]])], np.float64)
    p_mat = np.array([p2[i][j] for i in range(4) for j in range(2)], np.float64)
    return np.linalg.solve(a_mat, p_mat).astype(np.float32)     # (the reference solves in float32; float64 here)


def get_points(prob_map, img_h, img_w, settings, engine=None):
    """netutils.py:78-100 for a [1,H,W] probability map -> float64 [3,K]."""
    if prob_map.dim() == 2:
        prob_map = prob_map.unsqueeze(0)
    if prob_map.shape[0] != 1:
        raise ValueError("get_points takes one frame (the reference merges a batch: netutils.py:59-61)")
    e = engine or Engine(img_h, img_w, 1, prob_map.device.index or 0, settings.nms_dist,
                         settings.confidence_thresh, settings.border_remove, True)
    xy, conf, _, _ = e.get_points(prob_map)[0]
    return _points_array(xy, conf)


def get_descriptors(points, descriptors_map, img_h, img_w, settings, engine=None):
    """netutils.py:103-121 -> float32 [D,K]."""
    d = descriptors_map.shape[1]
    if points.shape[1] == 0:
        return np.zeros((d, 0))
    if descriptors_map.dim() != 4 or descriptors_map.shape[0] != 1:
        raise ValueError("get_descriptors takes one frame's map [1,D,H/8,W/8]")
    # the descriptor head is not needed to SAMPLE a map: an engine made here is detector-only (small workspace)
    e = engine or Engine(img_h, img_w, 1, descriptors_map.device.index or 0, settings.nms_dist,
                         settings.confidence_thresh, settings.border_remove, False)
    if e.desc_dim != d or e.h != img_h or e.w != img_w:
        raise ValueError("engine geometry %dx%d D=%d does not match the map (%dx%d, D=%d)" % (e.h, e.w, e.desc_dim, img_h, img_w, d))
    pts = np.ascontiguousarray(np.asarray(points, np.float64)[:2, :].T)       # [K,2] (x, y), float64 as in the reference
    return np.ascontiguousarray(e.sample_descriptors(descriptors_map[0], pts).T)


def get_features(frame, net):
    """python/src/inference.py:99-104: rows [x, y, confidence, descriptor(128)]."""
    points, descriptors = net.run(frame)
    return np.hstack((points.T, descriptors.T))


def get_best_correspondences(stop_features, features, engine):
    """python/src/inference.py:88-96 (cv2.BFMatcher(NORM_L2, crossCheck=True)) on the GPU:
    -> (rows of `features` that found a mutual nearest neighbour, their indices into `stop_features`)."""
    match, _ = engine.match(features[:, 3:].astype(np.float32), stop_features[:, 3:].astype(np.float32),
                            cross_check=True)
    keep = np.flatnonzero(match >= 0)
    return features[keep], match[keep]


def _points_array(xy, conf):
    pts = np.zeros((3, len(conf)))
    pts[0], pts[1], pts[2] = xy[:, 0], xy[:, 1], conf
    return pts


class InferenceWrapper:
    """python/src/inferencewrapper.py:12-46,70-81."""

    def __init__(self, weights_path, settings, device=0, max_batch=1):
        self.name = "SuperPoint"
        self.settings = settings
        self.net = SuperPoint(settings, device, max_batch)
        self.net.load_state_dict(load_checkpoint_for_inference(weights_path))

    def prepare_input(self, img):
        if not torch.is_tensor(img):
            assert img.ndim == 3
            assert img.dtype == np.float32, "Image must be float32."
            assert img.shape[2] == 3, "Image must be rgb."
            input_tensor = torch.from_numpy(img.copy().transpose((2, 0, 1))).unsqueeze(0)
        else:
            input_tensor = img
        return input_tensor

    def run(self, img):
        """-> (points float64 [3,K], descriptors float32 [D,K])."""
        x = self.prepare_input(img)
        if x.shape[0] != 1:
            raise ValueError("run() takes one frame; use run_batch() for several")
        pts, desc = self.run_batch(x)[0]
        return pts, desc

    def run_with_homography_adaptation(self, img, config, homographies=None, rng=None):
        """python/src/inferencewrapper.py:48-68: img [N,C,H,W] -> list of N point arrays [3,K] from the probability
        maps aggregated over 1 + config.num views.  `homographies` [num,8] fixes the views; otherwise they are drawn
        with sample_homography(rng)."""
        x = self.prepare_input(img)
        h, w = x.shape[2], x.shape[3]
        e = self.net.engine(h, w, x.shape[0])
        if homographies is None:
            homographies = np.stack([sample_homography((h, w), config, rng) for _ in range(config.num)])
        prob = e.homography_adaptation(x, homographies, None, config.valid_border_margin, config.aggregation)
        return [_points_array(xy, conf) for xy, conf, _, _ in e.get_points(prob)]

    def trace(self, img, out_file_name):
        """python/src/inferencewrapper.py:83-91.  The reference writes two files: a TorchScript trace (for its TRTorch
        path -- no meaning here, the kernels ARE the compiled model, so it is not written) and "just weights for cpp":
        the state dict as a flat {name: tensor} file `<out>_params.pt` with the first dotted component of every key
        dropped, exactly as the reference does (`'.'.join(k.split('.')[1:])`).  Returns the path written."""
        del img
        sd = self.net._state_dict
        flat = {".".join(k.split(".")[1:]): (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v)))
                for k, v in sd.items()}
        path = out_file_name + "_params.pt"
        torch.save(flat, path, _use_new_zipfile_serialization=True)
        return path

    def export_state_dict(self, path):
        """The loaded checkpoint as a flat {name: tensor} file with the FULL key names: the form
        `superpoint::SuperPoint(file, false)` of cpp/superpoint.hpp (and fpc_load_weights) take besides the trainer's
        checkpoint."""
        sd = self.net._state_dict
        torch.save({k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}, path)
        return path

    def run_batch(self, frames):
        """frames [n,3,H,W] -> list of (points [3,K], descriptors [D,K]), one per frame."""
        h, w = frames.shape[2], frames.shape[3]
        e = self.net.engine(h, w, frames.shape[0])
        out = []
        for xy, conf, desc, _ in e.detect(frames):
            d = desc.T.copy() if desc is not None else np.full((e.desc_dim, len(conf)), np.nan, np.float32)
            out.append((_points_array(xy, conf), d))
        return out
