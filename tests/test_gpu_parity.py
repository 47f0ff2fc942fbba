"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle
and the committed golden fixtures.  Need a real MI355X:  pytest -m gpu

Bars (BASELINE.json north_star): dense float outputs within 1e-4 of the reference;
keypoint indices after NMS identical.  Integer / index work (threshold, NMS, sort,
border crop) is compared bit-exact on identical inputs."""
import os

import numpy as np
import pytest

import fpc_amd  # noqa: F401
from fpc_amd import arch, synth

pytestmark = pytest.mark.gpu

SPEC = arch.state_dict_spec()
ATOL = 1e-4     # the tolerance north_star states


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def engine(h, w, b=1, **kw):
    from fpc_amd.engine import Engine
    return Engine(h, w, max_batch=b, **kw)


def oracle_mod():
    from oracle import oracle
    return oracle


def test_native_library_is_loaded(torch_gpu):
    from fpc_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libfpc.so" in maps


@pytest.mark.parametrize("dtype", ["f32", "f32_split", "f32_split_f16"])
def test_f1_small_frame_dense_maps(torch_gpu, golden_dir, dtype):
    g = np.load(os.path.join(golden_dir, "f1_layers_32x48.npz"))
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    frame = synth.make_batch(int(g["seed_frame"]), 1, 32, 48)
    e = engine(32, 48, dtype=dtype)
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(frame)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc.cpu().numpy(), g["desc_map"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(prob.cpu().numpy(), g["prob"], rtol=0, atol=ATOL)
    # and much tighter against the oracle, relative to the activation scale
    o_prob, o_desc, o_logits = oracle_mod().forward(frame, sd, SPEC)
    assert np.max(np.abs(logits.cpu().numpy() - o_logits)) < 2e-5 * max(1.0, np.abs(o_logits).max())
    assert np.max(np.abs(desc.cpu().numpy() - o_desc)) < 2e-5 * max(1.0, np.abs(o_desc).max())
    e.close()


def _check_frame_against_oracle_postproc(oracle, prob_b, desc_b, res, h, w):
    """HIP post-processing vs the oracle's on the SAME probability / descriptor maps."""
    xy, conf, d, ncand = res
    oxs, oys, oconf, oncand = oracle.get_points(prob_b)
    assert ncand == oncand
    np.testing.assert_array_equal(xy[:, 0], oxs)
    np.testing.assert_array_equal(xy[:, 1], oys)
    np.testing.assert_array_equal(conf, oconf)          # bit-exact: the same floats, re-ordered
    if d is not None and len(oxs):
        od = oracle.get_descriptors(desc_b, oxs, oys, h, w)
        np.testing.assert_allclose(d, od, rtol=0, atol=2e-6)
        np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, rtol=1e-5)


@pytest.mark.parametrize("dtype", ["f32", "f32_split", "f32_split_f16"])
def test_f1_per_layer_taps_on_the_device(torch_gpu, golden_dir, dtype):
    """Fixture F1 -- forward hooks on the reference's own modules -- against the tensors the HIP path leaves in its
    workspace (fpc_read_activation), layer by layer at the 1e-4 bar: a mid-network indexing error cannot cancel out
    before the final maps.  (The un-pooled stem output and a block's inner h never reach memory in the fused plan.)"""
    g = np.load(os.path.join(golden_dir, "f1_layers_32x48.npz"))
    h, w = int(g["h"]), int(g["w"])
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    frame = synth.make_batch(int(g["seed_frame"]), 1, h, w)
    from fpc_amd.engine import Engine
    # default plan (a one-frame call runs the latency variants: generation-2 Winograd on fine tiles); the batch plan's
    # kernels on the same frame (no_latency_tiles: the F(4x4,3x3) kernel, wblock36_mfma.h, on every 64- / 128-channel layer);
    # generation 2 forced; direct convolutions; unfused
    for kw in (dict(), dict(plan_flags=["no_latency_tiles"]), dict(plan_flags=["no_latency_tiles", "winograd_gen2"]),
               dict(plan_flags=["no_winograd"]), dict(plan_flags=["no_fused_blocks"])):
        if kw and dtype != "f32":
            continue
        e = engine(h, w, dtype=dtype, **kw)
        e.load_state_dict(sd)
        e.forward(frame)
        for name in Engine.ACTIVATIONS:
            got = e.activation(name).cpu().numpy()
            want = g["tap_" + name]
            assert got.shape == want.shape, (name, got.shape, want.shape)
            np.testing.assert_allclose(got, want, rtol=0, atol=ATOL, err_msg="%s %s %s" % (dtype, kw, name))
        e.close()


@pytest.mark.parametrize("dtype", ["f32", "f32_batch_plan", "f32_split", "f32_split_f16"])
@pytest.mark.parametrize("tag", ["qvga", "vga", "magicpoint_qvga"])
def test_f5_end_to_end(torch_gpu, golden_dir, tag, dtype):
    """The reference's own outputs (tests/golden/make_golden.py) at the north_star bar -- dense maps within
    1e-4, keypoint set identical -- for the fp32 MFMA path and for the split-operand path (block_x3.h).
    "f32_batch_plan": the kernels a 32-frame batch runs (Winograd F(4x4,3x3) blocks), on the fixture's one frame."""
    plan = dict(plan_flags=["no_latency_tiles"]) if dtype == "f32_batch_plan" else {}
    dtype = "f32" if dtype == "f32_batch_plan" else dtype
    g = np.load(os.path.join(golden_dir, "f5_e2e_%s.npz" % tag))
    h, w = int(g["h"]), int(g["w"])
    de = bool(int(g["descriptor_enabled"]))
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    frame = synth.make_batch(int(g["seed_frame"]), 1, h, w)
    e = engine(h, w, descriptor_enabled=de, dtype=dtype, **plan)
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(frame)
    prob, desc, logits = prob.cpu().numpy(), desc.cpu().numpy(), logits.cpu().numpy()
    # dense maps vs the reference's probes
    np.testing.assert_allclose(logits.ravel()[::7], g["logits_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc.ravel()[::11], g["desc_map_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(prob.ravel()[::13], g["prob_probe"], rtol=0, atol=ATOL)
    res = e.detect(frame)[0]
    oracle = oracle_mod()
    _check_frame_against_oracle_postproc(oracle, prob[0], desc[0], res, h, w)
    # keypoint SET identical to the reference's (fixtures are tie-safe: make_golden.py)
    xy, conf, d, ncand = res
    gx, gy, gc = g["points_x"].astype(np.int64), g["points_y"].astype(np.int64), g["points_conf"]
    assert ncand == int(g["n_candidates"])
    assert len(conf) == len(gc)
    np.testing.assert_array_equal(np.sort(xy[:, 1].astype(np.int64) * w + xy[:, 0]), np.sort(gy * w + gx))
    np.testing.assert_allclose(conf, gc, rtol=0, atol=ATOL)
    if de:
        order = {int(i): k for k, i in enumerate((gy * w + gx).tolist())}
        perm = np.array([order[int(i)] for i in xy[:, 1].astype(np.int64) * w + xy[:, 0]])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        np.testing.assert_allclose(d[inv[g["desc_subset_idx"]]], g["desc_subset"], rtol=0, atol=ATOL)
    else:
        assert d is None and not desc.any()
    e.close()


F3_CASES = ["empty", "below_thresh", "single", "single_border", "pair_d4", "pair_d5", "pair_diag4",
            "chain", "border_suppresses", "corners", "dense_cluster", "threshold_edge",
            "rand_64x96", "rand_120x160", "rand_full_48x64"]


def test_f3_get_points_fixtures(torch_gpu, golden_dir):
    """get_points alone (netutils.py:78-100) on the reference's hand-built maps: exact."""
    torch = torch_gpu
    g = np.load(os.path.join(golden_dir, "f3_get_points.npz"))
    engines = {}
    for name in F3_CASES:
        h, w = [int(v) for v in g[name + "_hw"]]
        pm = np.zeros(h * w, np.float32)
        pm[g[name + "_idx"]] = g[name + "_val"]
        ref = g[name + "_out"]
        if (h, w) not in engines:
            engines[(h, w)] = engine(h, w, descriptor_enabled=False)
        xy, conf, _, _ = engines[(h, w)].get_points(torch.from_numpy(pm.reshape(1, h, w)))[0]
        assert len(conf) == ref.shape[1], name
        np.testing.assert_array_equal(xy[:, 0], ref[0].astype(np.int32), err_msg=name)
        np.testing.assert_array_equal(xy[:, 1], ref[1].astype(np.int32), err_msg=name)
        np.testing.assert_array_equal(conf, ref[2].astype(np.float32), err_msg=name)
    for e in engines.values():
        e.close()


def test_get_points_ties_and_degenerate_maps(torch_gpu):
    """Cases the reference leaves unspecified or that stress the kernels: exact ties inside
    the NMS window, an all-pass map, a constant map -- against the oracle (same defined order)."""
    torch = torch_gpu
    oracle = oracle_mod()
    h, w = 64, 96
    e = engine(h, w, descriptor_enabled=False)
    rng = np.random.Generator(np.random.PCG64(4))
    maps = []
    m = np.zeros((h, w), np.float32)
    m[10, 10:40] = 0.5            # a run of exact ties: greedy keeps every 5th by index order
    m[20:50, 60] = 0.25
    maps.append(m)
    maps.append(np.full((h, w), 0.3, np.float32))                      # constant: all ties
    maps.append(rng.uniform(0.02, 1.0, (h, w)).astype(np.float32))     # all pass
    q = (rng.integers(0, 8, (h, w)) / 8.0).astype(np.float32)          # heavy ties, some below threshold
    maps.append(q)
    maps.append(np.zeros((h, w), np.float32))                          # nothing
    d = np.linspace(1.0, 0.02, w, dtype=np.float32)                    # monotone chains along rows (many rounds)
    maps.append(np.tile(d, (h, 1)))
    for i, m in enumerate(maps):
        xy, conf, _, ncand = e.get_points(torch.from_numpy(m[None]))[0]
        oxs, oys, oconf, oncand = oracle.get_points(m)
        assert ncand == oncand, i
        np.testing.assert_array_equal(xy[:, 0], oxs, err_msg=str(i))
        np.testing.assert_array_equal(xy[:, 1], oys, err_msg=str(i))
        np.testing.assert_array_equal(conf, oconf, err_msg=str(i))
    e.close()


def test_nms_radius_and_border_options(torch_gpu):
    """nms_dist / border_remove / conf_thresh are runtime settings (settings.py:3-8)."""
    torch = torch_gpu
    oracle = oracle_mod()
    h, w = 48, 64
    rng = np.random.Generator(np.random.PCG64(8))
    m = np.where(rng.uniform(0, 1, (h, w)) < 0.2, rng.uniform(0.0, 1.0, (h, w)), 0.0).astype(np.float32)
    for r, bw, thr in [(0, 0, 0.015), (1, 2, 0.3), (2, 0, 0.015), (7, 6, 0.1), (4, 4, 0.5)]:
        e = engine(h, w, descriptor_enabled=False, nms_dist=r, border_remove=bw, conf_thresh=thr)
        xy, conf, _, ncand = e.get_points(torch.from_numpy(m[None]))[0]
        oxs, oys, oconf, oncand = oracle.get_points(m, thr, r, bw)
        assert ncand == oncand
        np.testing.assert_array_equal(xy[:, 0], oxs)
        np.testing.assert_array_equal(xy[:, 1], oys)
        np.testing.assert_array_equal(conf, oconf)
        e.close()


def test_descriptor_sampling_against_oracle(torch_gpu):
    """get_descriptors (netutils.py:103-121) incl. the frame corners (zero-padding taps)."""
    torch = torch_gpu
    oracle = oracle_mod()
    h, w = 64, 96
    rng = np.random.Generator(np.random.PCG64(12))
    dm = rng.uniform(0.0, 2.0, (1, 128, h // 8, w // 8)).astype(np.float32)
    pm = np.zeros((1, h, w), np.float32)
    pts = [(0, 0), (w - 1, h - 1), (w - 1, 0), (0, h - 1), (8, 8), (16, 24), (33, 17), (95, 32), (48, 63), (5, 44)]
    for k, (x, y) in enumerate(pts):
        pm[0, y, x] = 0.9 - 0.01 * k
    e = engine(h, w, border_remove=0, nms_dist=2)
    xy, conf, d, _ = e.get_points(torch.from_numpy(pm), torch.from_numpy(dm))[0]
    assert len(conf) == len(pts)
    od = oracle.get_descriptors(dm[0], xy[:, 0], xy[:, 1], h, w)
    np.testing.assert_allclose(d, od, rtol=0, atol=2e-6)
    e.close()


def test_batch_equals_single_frames(torch_gpu):
    """Frames of a batch are independent (the reference façade is batch-1: netutils.py:59-61)."""
    h, w, n = 96, 128, 5
    sd = synth.make_state_dict(31, dustbin_bias=5.0)
    frames = synth.make_batch(40, n, h, w)
    eb = engine(h, w, n)
    eb.load_state_dict(sd)
    batch = eb.detect(frames)
    pb, db, lb = eb.forward(frames)
    e1 = engine(h, w, 1)
    e1.load_state_dict(sd)
    for i in range(n):
        xy, conf, d, nc = e1.detect(frames[i:i + 1])[0]
        bxy, bconf, bd, bnc = batch[i]
        assert nc == bnc
        np.testing.assert_array_equal(xy, bxy)
        np.testing.assert_array_equal(conf, bconf)
        np.testing.assert_array_equal(d, bd)
        p1, d1, l1 = e1.forward(frames[i:i + 1])
        np.testing.assert_array_equal(l1[0].cpu().numpy(), lb[i].cpu().numpy())
    # repeated calls are deterministic
    again = eb.detect(frames)
    for a, b in zip(batch, again):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[2], b[2])
    eb.close()
    e1.close()


def test_full_size_batch_properties(torch_gpu):
    """BASELINE.json configs[1] at full size (32 x 640x480) through size-independent properties:
    every frame's post-processing is exact against the oracle run on the HIP dense maps, keypoints
    respect the NMS radius and the border, confidences descend, descriptors have unit norm."""
    h, w, n = 480, 640, 32
    sd = synth.make_state_dict(0, dustbin_bias=7.0)
    frames = synth.make_batch(100, n, h, w)
    e = engine(h, w, n)
    e.load_state_dict(sd)
    prob, desc, _ = e.forward(frames)
    res = e.detect(frames)
    oracle = oracle_mod()
    for i in (0, 13, 31):
        _check_frame_against_oracle_postproc(oracle, prob[i].cpu().numpy(), desc[i].cpu().numpy(), res[i], h, w)
    for i in (7, 29):          # one frame of each sub-batch against the oracle's own forward
        o_prob, o_desc, o_logits = oracle.forward(frames[i:i + 1], sd, SPEC)
        assert np.max(np.abs(prob[i].cpu().numpy() - o_prob[0])) < ATOL
        assert np.max(np.abs(desc[i].cpu().numpy() - o_desc[0])) < ATOL
    for xy, conf, d, ncand in res:
        assert len(conf) > 100 and ncand >= len(conf)
        assert np.all(np.diff(conf) <= 0)
        assert xy[:, 0].min() >= 4 and xy[:, 0].max() < w - 4 and xy[:, 1].min() >= 4 and xy[:, 1].max() < h - 4
        grid = np.zeros((h, w), bool)
        grid[xy[:, 1], xy[:, 0]] = True
        # no two keypoints within infinity-distance 4: every 5x5-aligned window sum of a dilated grid
        ys, xs = xy[:, 1], xy[:, 0]
        for dy in range(-4, 5):
            for dx in range(-4, 5):
                if dy == 0 and dx == 0:
                    continue
                yy, xx = ys + dy, xs + dx
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                assert not grid[yy[ok], xx[ok]].any()
        np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, rtol=1e-5)
    e.close()


def test_reference_style_wrapper(torch_gpu, tmp_path):
    """InferenceWrapper(weights_path, settings).run(img) with a checkpoint FILE in the
    reference's layout (saveutils.py:57-62) -> (points [3,K] float64, descriptors [D,K])."""
    torch = torch_gpu
    from fpc_amd.inference import InferenceWrapper, SuperPointSettings
    sd = synth.make_state_dict(21, dustbin_bias=7.0)
    f = str(tmp_path / "super_point_0.pt")
    torch.save({"epoch": 0, "model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in sd.items()},
                "optimizer_state_dict": {}, "scaler_state_dict": {}}, f)
    wrap = InferenceWrapper(f, SuperPointSettings())
    img = synth.make_frame(300, 240, 320)
    points, desc = wrap.run(img)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f5_e2e_qvga.npz"))
    assert points.dtype == np.float64 and points.shape == (3, len(g["points_x"]))
    assert desc.shape == (128, points.shape[1]) and desc.dtype == np.float32
    np.testing.assert_array_equal(np.sort(points[1].astype(np.int64) * 320 + points[0].astype(np.int64)),
                                  np.sort(g["points_y"].astype(np.int64) * 320 + g["points_x"]))
    with pytest.raises(FileNotFoundError):
        InferenceWrapper(str(tmp_path / "missing.pt"), SuperPointSettings())
    # the same wrapper in the split-operand mode, and on the C++ network's flat dict (gray tensor in, 256-D out)
    st = SuperPointSettings()
    st.dtype = "f32_split_f16"
    p2, d2 = InferenceWrapper(f, st).run(img)
    np.testing.assert_array_equal(np.sort(p2[1].astype(np.int64) * 320 + p2[0].astype(np.int64)),
                                  np.sort(points[1].astype(np.int64) * 320 + points[0].astype(np.int64)))
    vf = str(tmp_path / "sp_vgg.pt")
    torch.save({k: torch.from_numpy(v.copy()) for k, v in synth.make_vgg_state_dict(32, 3.0).items()}, vf)
    gray = torch.from_numpy(synth.make_batch(401, 1, 240, 320, gray=True)[:, :1].copy())
    pv, dv = InferenceWrapper(vf, SuperPointSettings()).run(gray)
    assert dv.shape == (256, pv.shape[1]) and pv.shape[1] > 50


def test_error_codes(torch_gpu):
    from fpc_amd import _lib
    from fpc_amd.engine import Engine
    with pytest.raises(_lib.FpcError) as ei:
        Engine(100, 128)                      # not a multiple of 16
    assert ei.value.code == -1
    e = Engine(32, 48)
    with pytest.raises(_lib.FpcError) as ei:  # forward before weights
        e.forward(np.zeros((1, 3, 32, 48), np.float32))
    assert ei.value.code == -4
    sd = synth.make_state_dict(1)
    del sd["encoder.layer1.0.bn1.running_var"]
    with pytest.raises(_lib.FpcError) as ei:  # strict load (saveutils.py:10-14)
        e.load_state_dict(sd)
    assert ei.value.code == -5 and "encoder.layer1.0.bn1.running_var" in str(ei.value)
    sd = synth.make_state_dict(1)
    sd["detector.layer.0.conv1.weight"] = sd["detector.layer.0.conv1.weight"][:64]
    with pytest.raises(_lib.FpcError):
        e.load_state_dict(sd)
    e.close()


def test_cpp_entry_point(torch_gpu, tmp_path):
    """superpoint::SuperPoint(file, false).ProcessFrame(gray) (feature-point-cnn_amd/cpp/superpoint.hpp,
    mirroring cpp/src/superpoint.h:12-18) against the Python host on the same gray frame."""
    import subprocess
    torch = torch_gpu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    demo = os.path.join(root, "feature-point-cnn_amd", "lib", "fpc_demo")
    assert os.path.exists(demo), "run __graft_entry__.build()"
    h, w = 240, 320
    sd = synth.make_state_dict(21, dustbin_bias=7.0)
    ck = str(tmp_path / "super_point_0.pt")
    torch.save({"epoch": 0, "model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in sd.items()},
                "optimizer_state_dict": {}, "scaler_state_dict": {}}, ck)
    gray = synth.make_frame(300, h, w, gray=True)
    gray[..., 0].tofile(str(tmp_path / "frame.f32"))
    out = str(tmp_path / "pts.txt")
    msg = subprocess.check_output([demo, ck, str(tmp_path / "frame.f32"), str(h), str(w), out]).decode()
    got = np.loadtxt(out, ndmin=2)
    e = engine(h, w, in_channels=1)       # ProcessFrame(gray) uploads the single plane
    e.load_state_dict(sd)
    xy, conf, d, _ = e.detect(np.ascontiguousarray(gray[..., :1].transpose(2, 0, 1)[None]))[0]
    assert msg.startswith("%d feature points" % len(conf))
    np.testing.assert_array_equal(got[:, 0].astype(np.int32), xy[:, 0])
    np.testing.assert_array_equal(got[:, 1].astype(np.int32), xy[:, 1])
    np.testing.assert_allclose(got[:, 2], conf, rtol=1e-6)
    np.testing.assert_allclose(got[:, 3:6], d[:, 0:3], atol=1e-6)
    np.testing.assert_allclose(got[:, 6], d[:, 127], atol=1e-6)
    e.close()


@pytest.mark.parametrize("dtype", ["f32", "f32_split", "f32_split_f16"])
def test_hd_frames_and_odd_batch(torch_gpu, dtype):
    """BASELINE.json configs[4] geometry (1280x960) in fp32, and a batch that does not split evenly
    over the sub-batch streams: dense maps against the oracle, post-processing exact."""
    h, w, n = 960, 1280, 3
    sd = synth.make_state_dict(5, dustbin_bias=7.0)
    frames = synth.make_batch(700, n, h, w)
    e = engine(h, w, n, dtype=dtype)
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(frames)
    res = e.detect(frames)
    oracle = oracle_mod()
    o_prob, o_desc, o_logits = oracle.forward(frames[2:3], sd, SPEC)
    assert np.max(np.abs(logits[2].cpu().numpy() - o_logits[0])) < ATOL
    assert np.max(np.abs(desc[2].cpu().numpy() - o_desc[0])) < ATOL
    assert np.max(np.abs(prob[2].cpu().numpy() - o_prob[0])) < ATOL
    for i in range(n):
        _check_frame_against_oracle_postproc(oracle, prob[i].cpu().numpy(), desc[i].cpu().numpy(), res[i], h, w)
        assert len(res[i][1]) > 1000
    e.close()


def test_descriptor_matching(torch_gpu):
    """Next row: brute-force L2 matching with cross check (inference.py:88-96) and
    first-within-tolerance (cpp/src/main.cc:18-29) against the CPU restatement.  Parity for this row
    is unpinned (cv2 is not installed, no golden vectors); indices must agree exactly on tie-safe data."""
    oracle = oracle_mod()
    h, w = 240, 320
    sd = synth.make_state_dict(21, dustbin_bias=7.0)
    e = engine(h, w, 2)
    e.load_state_dict(sd)
    res = e.detect(synth.make_batch(300, 2, h, w))
    da, db = res[0][2], res[1][2]                       # real unit-norm descriptors of two frames
    assert len(da) > 300 and len(db) > 300
    rng = np.random.Generator(np.random.PCG64(3))
    db_near = (da[rng.permutation(len(da))[:400]] + rng.normal(0, 0.02, (400, 128))).astype(np.float32)  # true matches
    db_near /= np.linalg.norm(db_near, axis=1, keepdims=True)
    for q, t in ((da, db), (da, db_near), (db_near, da), (da[:1], db), (da[:130], db[:129])):
        for cross, md in ((True, 0.0), (False, 0.0), (True, 0.7)):
            m, d = e.match(q, t, cross, md)
            om, od = oracle.match(q, t, cross, md)
            # real descriptors are not guaranteed tie-free: where the index differs, the two candidates must be
            # equidistant to within the tolerance the distances themselves are held to (checked in double; the
            # kernel forms |q|^2 + |t|^2 - 2 q.t in fp32, a few ulps of 2.0 on d^2)
            for i in np.nonzero(m != om)[0]:
                assert m[i] >= 0 and om[i] >= 0
                dd = np.linalg.norm(q[i].astype(np.float64) - t[[m[i], om[i]]].astype(np.float64), axis=1)
                assert abs(dd[0] - dd[1]) < 2e-5, (i, m[i], om[i], dd)
            assert (m != om).sum() <= 2
            np.testing.assert_allclose(d, od, rtol=0, atol=2e-5)
        np.testing.assert_array_equal(e.first_within(q, t, 0.8), oracle.first_within(q, t, 0.8))
        np.testing.assert_array_equal(e.first_within(q, t, 0.3), oracle.first_within(q, t, 0.3))
    m, _ = e.match(da, db_near, True)
    assert (m >= 0).sum() >= 100                        # planted pairs survive the cross check
    m, _ = e.match(da, np.zeros((0, 128), np.float32))
    assert (m == -1).all()
    from fpc_amd.inference import get_best_correspondences
    fa = np.hstack((np.zeros((len(da), 3)), da))
    fb = np.hstack((np.zeros((len(db_near), 3)), db_near))
    corr, idx = get_best_correspondences(fb, fa, e)
    assert len(corr) == len(idx) and len(idx) >= 100
    e.close()


def test_packed_weight_blob_round_trip(torch_gpu):
    """The blob rank 0 broadcasts: the zero-copy tensor view aliases the library's buffer, and an
    engine that only ever received the blob computes the same results."""
    torch = torch_gpu
    h, w = 64, 96
    sd = synth.make_state_dict(9, dustbin_bias=4.0)
    frame = synth.make_batch(11, 1, h, w)
    a = engine(h, w)
    a.load_state_dict(sd)
    blob = a.export_packed()
    view = a.packed_view()
    assert view.dtype == torch.uint8 and view.numel() == a.packed_size() == blob.nbytes
    np.testing.assert_array_equal(view.cpu().numpy(), blob)
    b = engine(h, w)
    vb = b.packed_view()
    vb.copy_(view)                       # what dist.broadcast does on the receiving ranks
    torch.cuda.synchronize()
    b.mark_weights_loaded()
    ra, rb = a.detect(frame)[0], b.detect(frame)[0]
    np.testing.assert_array_equal(ra[0], rb[0])
    np.testing.assert_array_equal(ra[2], rb[2])
    c = engine(h, w)
    c.import_packed(blob)
    np.testing.assert_array_equal(c.detect(frame)[0][2], ra[2])
    for e in (a, b, c):
        e.close()


@pytest.mark.parametrize("env", [
    {"FPC_WINOGRAD": "0"},                                   # direct fused blocks everywhere
    {"FPC_WINOGRAD_DET": "0"},                               # Winograd encoder/descriptor, direct detector
    {"FPC_FUSE": "0", "FPC_FUSE_STEM": "0"},                 # one launch per convolution, separate max-pool
    {"FPC_STREAMS": "1", "FPC_SPLIT_HEADS": "0"},            # everything on one stream
    {"FPC_STREAMS": "3", "FPC_NMS_PASSES": "0"},             # three sub-batches; NMS finished by the sort kernel alone
    {"FPC_SPLIT_HEADS": "1"},                                # detector head + NMS on a side stream of its own
    {"FPC_NMS_ASIDE": "0"},                                  # NMS in line instead of beside the descriptor head
    {"FPC_NMS_CHUNKED": "0"},                                # survivors sorted by one workgroup per frame (round 1's kernel)
    {"FPC_NMS_CHUNKED": "0", "FPC_NMS_PASSES": "0"},         # ... which then also runs every round itself
    {"FPC_XCD_ORDER": "0", "FPC_MIN_SUB": "4"},              # plain tile order in the Winograd kernel; sub-batches of 4
    {"FPC_PERSIST_MIN": "0"},                                # Winograd kernel with one workgroup per tile
    {"FPC_WINOGRAD_IN1": "0"},                               # layer_in.1 as one fused direct block (no conv-only Winograd)
    {"FPC_LATENCY_TILES": "0"},                              # the batch plan's kernels (Winograd F(4x4,3x3)) on a 13-frame call
    {"FPC_LATENCY_TILES": "0", "FPC_XCD_ORDER": "0"},        # ... with the plain tile order
    {"FPC_LATENCY_TILES": "0", "FPC_PERSIST_MIN": "0"},      # ... and with the non-persistent grid size
    {"FPC_LATENCY_TILES": "0", "FPC_WINOGRAD_GEN": "2"},     # round 2's F(2x2,3x3) kernel
    {"FPC_LATENCY_TILES": "0", "FPC_WINOGRAD_GEN": "1"},     # round 1's
    {"FPC_LATENCY_TILES": "0", "FPC_WINOGRAD_DET_GEN": "1"},  # F(4x4,3x3) blocks, but the detector's 65 channels on round 1's kernel (FPC_PLAN_DETECTOR_GEN1)
    {"FPC_LATENCY_TILES": "0", "FPC_W36_PAIRED": "0"},       # the 64-channel layers on round 3's one-wave-per-SIMD kernel (FPC_PLAN_W36_ONE_WAVE)
    {"FPC_CONV_LEAN": "0", "FPC_STEM_LEAN": "0"},            # round 1's conv_mfma_kernel / round 3's stem where round 5's leaner instances apply
    {"FPC_LATENCY_TILES": "0", "FPC_CONV_LEAN": "0"},        # ... in the batch plan
])
def test_alternative_plans_agree(torch_gpu, golden_dir, env):
    """Every launch plan the library can be switched to (environment knobs read at fpc_create) must
    meet the same bar: dense maps within 1e-4 of the reference's probes, keypoint set identical."""
    g = np.load(os.path.join(golden_dir, "f5_e2e_qvga.npz"))
    h, w = int(g["h"]), int(g["w"])
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    frames = np.repeat(synth.make_batch(int(g["seed_frame"]), 1, h, w), 13, axis=0)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = engine(h, w, 13)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(frames)
    np.testing.assert_allclose(logits[12].cpu().numpy().ravel()[::7], g["logits_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc[5].cpu().numpy().ravel()[::11], g["desc_map_probe"], rtol=0, atol=ATOL)
    res = e.detect(frames)
    gx, gy = g["points_x"].astype(np.int64), g["points_y"].astype(np.int64)
    for i in (0, 6, 12):
        xy, conf, d, ncand = res[i]
        assert ncand == int(g["n_candidates"])
        np.testing.assert_array_equal(np.sort(xy[:, 1].astype(np.int64) * w + xy[:, 0]), np.sort(gy * w + gx))
    e.close()


def test_gray_frames_equal_replicated_rgb(torch_gpu):
    """in_channels = 1: a gray frame [n,1,H,W] gives what the reference computes after replicating the
    plane over 3 channels (dataset_utils.py:19-20) -- the stem filters are summed over the input channels."""
    h, w, n = 240, 320, 3
    sd = synth.make_state_dict(21, dustbin_bias=7.0)
    rgb = np.stack([synth.make_frame(300 + i, h, w, gray=True).transpose(2, 0, 1) for i in range(n)])
    gray = np.ascontiguousarray(rgb[:, :1])
    e3 = engine(h, w, n)
    e1 = engine(h, w, n, in_channels=1)
    e3.load_state_dict(sd)
    e1.load_state_dict(sd)
    p3, d3, l3 = e3.forward(rgb)
    p1, d1, l1 = e1.forward(gray)
    assert float((l3 - l1).abs().max()) < 2e-5 and float((d3 - d1).abs().max()) < 2e-5
    o_prob, o_desc, o_logits = oracle_mod().forward(rgb[:1], sd, SPEC)
    assert np.max(np.abs(l1[0].cpu().numpy() - o_logits[0])) < ATOL
    r3, r1 = e3.detect(rgb), e1.detect(gray)
    for a, b in zip(r3, r1):
        np.testing.assert_array_equal(np.sort(a[0][:, 1] * w + a[0][:, 0]), np.sort(b[0][:, 1] * w + b[0][:, 0]))
    with pytest.raises(ValueError):
        e1.forward(rgb)
    e3.close()
    e1.close()


# bf16 path (BASELINE.json configs[4]): bf16 activations / weights, fp32 accumulation.  The 1e-4 bound of
# north_star is an fp32 statement; bf16 storage has 8 significant bits, so the dense maps are held to the
# error that rounding predicts (about 2^-8 relative per tensor, a handful of tensors deep) and the sparse
# outputs to agreement with the fp32 path.  Post-processing runs in fp32 on whatever dense maps it is given
# and stays EXACT against the oracle on those maps.
BF16_LOGIT_MAX, BF16_LOGIT_RMS, BF16_KP_OVERLAP, BF16_DESC_COS = 0.25, 0.02, 0.97, 0.999
def _overlap_and_cosine(ra, rb):
    (xa, _, fa, _), (xb, _, fb, _) = ra, rb
    ia = {tuple(p): k for k, p in enumerate(xa.tolist())}
    ib = {tuple(p): k for k, p in enumerate(xb.tolist())}
    common = [p for p in ia if p in ib]
    cos = [float(np.dot(fa[ia[p]], fb[ib[p]])) for p in common] if fa is not None else [1.0]
    return len(common) / max(1, max(len(ia), len(ib))), min(cos) if cos else 1.0


# ... and, so that a mis-rounded or mis-indexed layer cannot hide inside that allowance, EVERY LAYER is also checked on
# its own (round-1 VERDICT): the layer's input tensor is read back from the device (fpc_read_activation), the oracle
# computes that one layer with bf16's roundings put where the mode puts them (oracle.bf16_emulated_layer: bf16 inputs,
# BN-folded bf16 weights, bf16 h, bf16 or fp32 output, double accumulation), and the device's output tensor must agree
# up to what is left: fp32-vs-double accumulation order, plus -- for a value that lands within that distance of a bf16
# rounding boundary -- one bf16 ulp of the output or of an h feeding it.  Nothing compounds across layers, so the bound
# is a per-element one: |got - want| <= 1.25 * 2^-7 * max(|want|, 1) everywhere (one ulp of a bf16 output is between 2^-8
# and 2^-7 of its value; an h flip moves the sum by at most 2^-7 |h| |w|; two ulps would be >= 2^-7), and at most 0.5 % of
# the elements may differ by more than 1e-4 * max(|want|, 1) at all.  Measured on the MI355X (VGA and 64 x HD): max
# 0.0078 = one ulp, 0.001 - 0.10 % of a layer's elements flipped.
BF16_LAYER_REL, BF16_LAYER_EXACT, BF16_LAYER_FLIPS = 1.25 * 2.0 ** -7, 1e-4, 0.005


def _check_bf16_layers(oracle, e16, frames, sd, frame_idx, what):
    """frames must be the batch of the engine's LAST forward call."""
    stem = oracle.bf16_emulated_stem(frames[frame_idx], sd)
    got0 = np.concatenate([e16.activation("pool", i, 1).cpu().numpy() for i in frame_idx], 0)
    # stored as bf16 (max-pool and round-to-nearest commute): the rounded emulation, up to an ulp where the order of the
    # fp32 accumulation decided the rounding
    want0 = oracle.round_bf16(stem)
    rel0 = np.abs(got0.astype(np.float64) - want0) / np.maximum(np.abs(want0), 1.0)
    assert float(rel0.max()) < BF16_LAYER_REL, (what + ": pooled stem output", float(rel0.max()))
    assert float((rel0 > BF16_LAYER_EXACT).mean()) < BF16_LAYER_FLIPS, what + ": pooled stem output"
    worst = 0.0
    for name, ins in oracle.BF16_LAYERS:
        if not e16.descriptor_enabled and (name.startswith("desc") or name == "up"):
            continue
        xin = [np.concatenate([e16.activation(t, i, 1).cpu().numpy() for i in frame_idx], 0) for t in ins]
        got = np.concatenate([e16.activation(name, i, 1).cpu().numpy() for i in frame_idx], 0)
        want = oracle.bf16_emulated_layer(name, xin, sd)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
        frac = float((rel > BF16_LAYER_EXACT).mean())
        worst = max(worst, float(rel.max()))
        print("bf16 layer %-10s %s: max rel %.3g, %.4f %% of elements off by more than %g" % (name, what, rel.max(), 100 * frac, BF16_LAYER_EXACT))
        assert float(rel.max()) < BF16_LAYER_REL, (what, name, float(rel.max()))
        assert frac < BF16_LAYER_FLIPS, (what, name, frac)
    return worst


@pytest.mark.gpu
def test_bf16_path_against_fp32_and_oracle(torch_gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "f5_e2e_vga.npz"))
    h, w = int(g["h"]), int(g["w"])
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    frame = synth.make_batch(int(g["seed_frame"]), 1, h, w)
    e32 = engine(h, w)
    e32.load_state_dict(sd)
    e16 = engine(h, w, dtype="bf16")
    e16.load_state_dict(sd)
    p32, d32, l32 = e32.forward(frame)
    p16, d16, l16 = e16.forward(frame)
    diff = (l32 - l16).float()
    assert float(diff.abs().max()) < BF16_LOGIT_MAX and float(diff.pow(2).mean().sqrt()) < BF16_LOGIT_RMS
    # against the reference's own probes of the dense maps (golden fixture), same bound
    assert np.max(np.abs(l16.cpu().numpy().ravel()[::7] - g["logits_probe"])) < BF16_LOGIT_MAX
    assert np.max(np.abs(d16.cpu().numpy().ravel()[::11] - g["desc_map_probe"])) < BF16_LOGIT_MAX
    _check_bf16_layers(oracle_mod(), e16, frame, sd, [0], "VGA")     # e16's last forward was on `frame`
    r32, r16 = e32.detect(frame)[0], e16.detect(frame)[0]
    # post-processing is exact on the bf16 engine's own dense maps (fpc_detect's detector block does the exp-softmax itself,
    # bit for bit what softmax_d2s_kernel makes of the logits fpc_forward returned)
    _check_frame_against_oracle_postproc(oracle_mod(), p16[0].cpu().numpy(), d16[0].cpu().numpy(), r16, h, w)
    e16u = engine(h, w, dtype="bf16", plan_flags=["no_fused_softmax"])
    e16u.load_state_dict(sd)
    for got, want in zip(e16u.detect(frame)[0], r16):
        np.testing.assert_array_equal(got, want)
    e16u.close()
    ov, cos = _overlap_and_cosine(r32, r16)
    assert ov >= BF16_KP_OVERLAP and cos >= BF16_DESC_COS
    # and against the reference's keypoints
    gset = set(zip(g["points_x"].astype(int).tolist(), g["points_y"].astype(int).tolist()))
    kset = {tuple(p) for p in r16[0].tolist()}
    assert len(gset & kset) / max(len(gset), len(kset)) >= BF16_KP_OVERLAP
    e32.close()
    e16.close()


@pytest.mark.gpu
def test_bf16_hd_batch_properties(torch_gpu):
    """BASELINE.json configs[4] AT ITS STATED SIZE: 64 frames of 1280x960 in bf16, one call.  Dense maps of three
    frames against the bf16-emulating oracle; post-processing exact against the oracle on three frames; batch == single
    frame bit for bit on two; NMS radius / border / ordering / unit-norm properties on all 64; MagicPoint variant."""
    h, w, n = 960, 1280, 64
    sd = synth.make_state_dict(5, dustbin_bias=7.0)
    # 8 distinct frames, repeated with a shift so that every slot of the batch holds a different image
    base = synth.make_batch(700, 8, h, w)
    frames = np.concatenate([np.roll(base, 16 * k, axis=3) for k in range(n // 8)], 0)
    e = engine(h, w, n, dtype="bf16")
    e.load_state_dict(sd)
    res = e.detect(frames)
    prob, desc, _ = e.forward(frames)
    oracle = oracle_mod()
    for i in (0, 29, n - 1):
        _check_frame_against_oracle_postproc(oracle, prob[i].cpu().numpy(), desc[i].cpu().numpy(), res[i], h, w)
    # every layer of three frames of the batch (first, one at the sub-batch boundary, last) against the bf16-emulating
    # oracle, each on the device's own input tensor.  The last call is fpc_forward: fpc_detect in this mode writes no
    # logits (fused softmax epilogue), so the "det.1" tap exists only after a forward (test_logits_tap_after_...)
    _check_bf16_layers(oracle, e, frames, sd, [0, 31, n - 1], "HD x64")
    for i in (3, 40):
        one = e.detect(frames[i:i + 1])[0]
        np.testing.assert_array_equal(one[0], res[i][0])
        np.testing.assert_array_equal(one[1], res[i][1])
        np.testing.assert_array_equal(one[2], res[i][2])
    for xy, conf, d, ncand in res:
        assert len(conf) > 1000 and np.all(np.diff(conf) <= 0) and ncand >= len(conf)
        assert xy[:, 0].min() >= 4 and xy[:, 0].max() < w - 4 and xy[:, 1].min() >= 4 and xy[:, 1].max() < h - 4
        grid = np.zeros((h + 8, w + 8), bool)
        grid[xy[:, 1] + 4, xy[:, 0] + 4] = True
        for dy in range(-4, 5):           # no two keypoints within infinity-distance 4 of each other
            for dx in range(-4, 5):
                if dy or dx:
                    assert not grid[xy[:, 1] + 4 + dy, xy[:, 0] + 4 + dx].any()
        np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, rtol=1e-5)
    e.close()
    m = engine(h, w, 1, dtype="bf16", descriptor_enabled=False)
    m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("descriptor.")})
    xy, conf, d, _ = m.detect(frames[:1])[0]
    assert d is None
    np.testing.assert_array_equal(xy, res[0][0])
    m.close()


def test_split_operand_path_equals_fp32_path_keypoints(torch_gpu):
    """dtype="f32_split" (fp32 tensors, products as six bf16 MFMAs on exactly split operands) on a full
    VGA batch: the same keypoints, in the same order, as the fp32 MFMA path on every frame, confidences and
    descriptors within the 1e-4 bar, and an exact oracle check of its post-processing."""
    h, w, n = 480, 640, 8
    sd = synth.make_state_dict(0, dustbin_bias=7.0)
    frames = synth.make_batch(100, n, h, w)
    a = engine(h, w, n)
    a.load_state_dict(sd)
    b = engine(h, w, n, dtype="f32_split")
    b.load_state_dict(sd)
    ra, rb = a.detect(frames), b.detect(frames)
    pb, db, lb = b.forward(frames)
    pa, da, la = a.forward(frames)
    assert float((la - lb).abs().max()) < ATOL and float((da - db).abs().max()) < ATOL
    oracle = oracle_mod()
    _check_frame_against_oracle_postproc(oracle, pb[5].cpu().numpy(), db[5].cpu().numpy(), rb[5], h, w)
    for (xa, ca, fa, na), (xb, cb, fb, nb) in zip(ra, rb):
        # identical keypoint set; the order may differ only between confidences closer than the bar
        ka, kb = xa[:, 1].astype(np.int64) * w + xa[:, 0], xb[:, 1].astype(np.int64) * w + xb[:, 0]
        oa, ob = np.argsort(ka), np.argsort(kb)
        np.testing.assert_array_equal(ka[oa], kb[ob])
        np.testing.assert_allclose(ca[oa], cb[ob], rtol=0, atol=ATOL)
        np.testing.assert_allclose(fa[oa], fb[ob], rtol=0, atol=ATOL)
        np.testing.assert_allclose(ca, cb, rtol=0, atol=ATOL)      # both descending
        assert abs(na - nb) <= 2                                   # candidates within 1e-4 of the threshold
    a.close()
    b.close()


def test_u8_frames_on_device(torch_gpu, golden_dir):
    """fpc_detect_u8: the conversion kernel is bit-exact against the oracle (itself pinned by F6 for the
    layouts the Python reference uses) on every byte value and on random frames, and the keypoints equal
    those of fpc_detect on the converted float frames."""
    oracle = oracle_mod()
    h, w, n = 64, 96, 3
    rng = np.random.Generator(np.random.PCG64(9))
    sd = synth.make_state_dict(3, dustbin_bias=2.0)
    col = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    col[0].reshape(-1)[:256 * 3] = np.repeat(np.arange(256, dtype=np.uint8), 3)   # every byte value in every channel
    e3 = engine(h, w, n)
    e3.load_state_dict(sd)
    for layout, code in (("rgb_hwc", 1), ("bgr_hwc", 2)):
        res = e3.detect_u8(col, layout)
        want = oracle.u8_to_float(col, code)
        np.testing.assert_array_equal(e3.u8_staging(n).cpu().numpy(), want)
        ref = e3.detect(want)
        for a, b in zip(res, ref):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_array_equal(a[2], b[2])
    g = np.load(os.path.join(golden_dir, "f6_u8_to_float.npz"))
    e3.detect_u8(np.broadcast_to(np.arange(256, dtype=np.uint8).repeat(3 * 24).reshape(1, 64, 96, 3), (1, 64, 96, 3)), "rgb_hwc")
    got = e3.u8_staging(1).cpu().numpy()[0, 0].ravel()[::24]
    np.testing.assert_array_equal(got[:256], g["table"])           # the reference's own table
    e3.close()
    e1 = engine(h, w, n, in_channels=1)
    e1.load_state_dict(sd)
    gray = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
    res = e1.detect_u8(gray, "gray")
    want = oracle.u8_to_float(gray, 0)
    np.testing.assert_array_equal(e1.u8_staging(n).cpu().numpy(), want)
    ref = e1.detect(want)
    for a, b in zip(res, ref):
        np.testing.assert_array_equal(a[0], b[0])
    e1.detect_u8(col, "bgr_hwc_gray")                               # cpp/src/camera.cc:17-18 (unpinned restatement)
    np.testing.assert_array_equal(e1.u8_staging(n).cpu().numpy(), oracle.u8_to_float(col, 3))
    with pytest.raises(Exception):
        e1.detect_u8(col, "rgb_hwc")                                # colour layout into a gray ctx
    e1.close()


# ---- the reference's C++ network (superpoint::SPModel, cpp/src/model.cc): arch="vgg" -------------------------
def _gray(seed, n, h, w):
    return synth.make_batch(seed, n, h, w, gray=True)[:, :1].copy()


@pytest.mark.parametrize("dtype", ["f32", "f32_split", "f32_split_f16"])
def test_vgg_network_against_reference_binary_fixtures(torch_gpu, golden_dir, dtype):
    """Dense maps of the HIP path vs the outputs of the reference's own cpp/src/model.cc (fixtures F7, written by
    oracle/_ref/ref_vgg_forward) at the 1e-4 bar, and vs the oracle; post-processing exact on those maps."""
    vspec = arch.vgg_state_dict_spec()
    g = np.load(os.path.join(golden_dir, "f7_vgg_32x48.npz"))
    sd = synth.make_vgg_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    fr = _gray(int(g["seed_frame"]), 1, 32, 48)
    e = engine(32, 48, in_channels=1, arch="vgg", dtype=dtype)
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(fr)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc.cpu().numpy(), g["desc"], rtol=0, atol=ATOL)
    e.close()
    g = np.load(os.path.join(golden_dir, "f7_vgg_qvga.npz"))
    h, w = int(g["h"]), int(g["w"])
    sd = synth.make_vgg_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    fr = _gray(int(g["seed_frame"]), 1, h, w)
    e = engine(h, w, in_channels=1, arch="vgg", dtype=dtype)
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(fr)
    prob, desc, logits = prob.cpu().numpy(), desc.cpu().numpy(), logits.cpu().numpy()
    np.testing.assert_allclose(logits.ravel()[::7], g["logits_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc.ravel()[::11], g["desc_probe"], rtol=0, atol=ATOL)
    oracle = oracle_mod()
    o_prob, o_desc, o_logits = oracle.vgg_forward(fr, sd, vspec)
    assert np.max(np.abs(logits - o_logits)) < 5e-5 and np.max(np.abs(desc - o_desc)) < 5e-6
    assert np.max(np.abs(prob - o_prob)) < 1e-5
    res = e.detect(fr)[0]
    assert res[2].shape[1] == 256 and len(res[1]) > 50
    _check_frame_against_oracle_postproc(oracle, prob[0], desc[0], res, h, w)
    e.close()


def test_vgg_batch_and_matching(torch_gpu):
    """C++ network on a VGA batch that splits over the sub-batch streams: batch == single frames, NMS / order /
    unit-norm properties, 256-D descriptor matching against the oracle."""
    h, w, n = 480, 640, 5
    sd = synth.make_vgg_state_dict(8, 3.0)
    fr = _gray(900, n, h, w)
    e = engine(h, w, n, in_channels=1, arch="vgg")
    e.load_state_dict(sd)
    res = e.detect(fr)
    one = e.detect(fr[2:3])[0]
    np.testing.assert_array_equal(one[0], res[2][0])
    np.testing.assert_array_equal(one[1], res[2][1])
    np.testing.assert_array_equal(one[2], res[2][2])
    for xy, conf, d, ncand in res:
        assert len(conf) > 100 and np.all(np.diff(conf) <= 0) and d.shape[1] == 256
        np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, rtol=1e-5)
    oracle = oracle_mod()
    q, t = res[0][2][:300], res[1][2][:400]
    m, dist = e.match(q, t, cross_check=True)
    om, od = oracle.match(q, t, True)
    np.testing.assert_array_equal(m, om)
    e.close()
    # detector only (descriptor head skipped) and 8-bit gray input give the same keypoints
    m = engine(h, w, 1, in_channels=1, arch="vgg", descriptor_enabled=False)
    m.load_state_dict(sd)
    xy, conf, d, _ = m.detect(fr[:1])[0]
    assert d is None
    np.testing.assert_array_equal(xy, res[0][0])
    u8 = np.clip(np.rint(fr[:1, 0] * 255.0), 0, 255).astype(np.uint8)
    a = m.detect_u8(u8, "gray")[0]
    b = m.detect(oracle.u8_to_float(u8, 0))[0]
    np.testing.assert_array_equal(a[0], b[0])
    m.close()
    with pytest.raises(Exception):
        engine(h, w, 1, in_channels=3, arch="vgg")        # the C++ network takes one gray plane


def test_cpp_entry_point_with_the_cpp_network(torch_gpu, tmp_path):
    """The reference's C++ frontend end to end on ITS network: superpoint::SuperPoint(file, false) with the flat
    {name: tensor} dict cpp/src/superpoint.cc:27-55 loads (torch.save of a dict, read by pt_reader.hpp without
    libtorch), ProcessFrame(gray) -> FeaturePoints with 256-float descriptors (torchutis.h:11-18)."""
    import subprocess
    torch = torch_gpu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    demo = os.path.join(root, "feature-point-cnn_amd", "lib", "fpc_demo")
    assert os.path.exists(demo), "run __graft_entry__.build()"
    h, w = 240, 320
    sd = synth.make_vgg_state_dict(32, 3.0)
    ck = str(tmp_path / "sp_vgg_flat.pt")
    torch.save({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, ck)
    gray = _gray(401, 1, h, w)
    gray[0, 0].tofile(str(tmp_path / "frame.f32"))
    out = str(tmp_path / "pts.txt")
    msg = subprocess.check_output([demo, ck, str(tmp_path / "frame.f32"), str(h), str(w), out]).decode()
    got = np.loadtxt(out, ndmin=2)
    e = engine(h, w, in_channels=1, arch="vgg")
    e.load_state_dict(sd)
    xy, conf, d, _ = e.detect(gray)[0]
    assert msg.startswith("%d feature points" % len(conf)) and len(conf) > 50
    np.testing.assert_array_equal(got[:, 0].astype(np.int32), xy[:, 0])
    np.testing.assert_array_equal(got[:, 1].astype(np.int32), xy[:, 1])
    np.testing.assert_allclose(got[:, 2], conf, rtol=1e-6)
    np.testing.assert_allclose(got[:, 3:6], d[:, 0:3], atol=1e-6)
    np.testing.assert_allclose(got[:, 7], d[:, 255], atol=1e-6)
    e.close()


def test_fp16_split_mode_reports_values_outside_its_range(torch_gpu):
    """dtype="f32_split_f16" splits operands into fp16 terms: a folded weight beyond 65504 is refused at load,
    an activation beyond it is reported (FPC_E_RANGE) when the results are fetched -- never silently wrong; the
    bf16-term mode takes the same checkpoints."""
    from fpc_amd._lib import FpcError
    h, w = 64, 96
    fr = synth.make_batch(5, 1, h, w)
    sd = synth.make_state_dict(3, dustbin_bias=2.0)
    big_w = dict(sd)
    big_w["encoder.layer2.1.conv1.weight"] = sd["encoder.layer2.1.conv1.weight"] * np.float32(4e6)
    e = engine(h, w, dtype="f32_split_f16")
    with pytest.raises(FpcError) as ei:
        e.load_state_dict(big_w)
    assert ei.value.code == -8
    big_a = dict(sd)                                   # moderate weights, activations growing x600 per block
    for k in ("encoder.layer1.0.bn2.weight", "encoder.layer1.0.identity_downsample.1.weight", "encoder.layer1.1.bn2.weight"):
        big_a[k] = sd[k] * np.float32(600.0)
    e.load_state_dict(big_a)
    with pytest.raises(FpcError) as ei:
        e.detect(fr)
    assert ei.value.code == -8
    e.load_state_dict(sd)                              # the flag is per call: a sane checkpoint works again
    assert len(e.detect(fr)[0][1]) > 0
    e.close()
    b = engine(h, w, dtype="f32_split")                # bf16 terms carry fp32's exponent range
    b.load_state_dict(big_a)
    b.detect(fr)
    b.close()


def test_f3b_random_maps_and_settings_on_gpu(torch_gpu, golden_dir):
    """Fixture F3b (the reference's get_points on 48 random maps with varying nms_dist / border_remove /
    confidence_thresh) through fpc_get_points: indices and confidence bits identical to the reference's."""
    g = np.load(os.path.join(golden_dir, "f3b_get_points_random.npz"))
    for i in range(48):
        k = "c%02d" % i
        h, w = [int(v) for v in g[k + "_hw"]]
        pm = np.zeros(h * w, np.float32)
        pm[g[k + "_idx"]] = g[k + "_val"]
        ref = g[k + "_out"]
        e = engine(h, w, 1, descriptor_enabled=False, nms_dist=int(g[k + "_par"][0]), border_remove=int(g[k + "_par"][1]),
                   conf_thresh=float(g[k + "_thr"]))
        xy, conf, d, _ = e.get_points(torch_gpu.from_numpy(pm.reshape(1, h, w)))[0]
        assert len(conf) == ref.shape[1], k
        np.testing.assert_array_equal(xy[:, 0], ref[0].astype(np.int32), err_msg=k)
        np.testing.assert_array_equal(xy[:, 1], ref[1].astype(np.int32), err_msg=k)
        np.testing.assert_array_equal(conf, ref[2].astype(np.float32), err_msg=k)
        e.close()


def test_f4b_descriptor_sampling_geometries_on_gpu(torch_gpu, golden_dir):
    """Fixture F4b through fpc_get_points' descriptor sampling: the reference's get_descriptors on wide / tall maps.
    The keypoints are planted in a probability map (distinct confidences, NMS off) so that the kernel samples
    exactly the fixture's points."""
    g = np.load(os.path.join(golden_dir, "f4b_get_descriptors_random.npz"))
    for i in range(5):
        h, w = [int(v) for v in g["c%d_hw" % i]]
        if h % 16 or w % 16:
            continue        # a full ctx needs multiples of 16 (descriptor head); the oracle test covers these cases
        pts, ref = g["c%d_pts" % i], g["c%d_out" % i]
        flat = pts[1].astype(np.int64) * w + pts[0]
        uniq, first = np.unique(flat, return_index=True)          # a pixel can only be one keypoint
        pm = np.zeros(h * w, np.float32)
        pm[uniq] = np.linspace(0.9, 0.1, len(uniq)).astype(np.float32)
        e = engine(h, w, 1, nms_dist=0, border_remove=0, conf_thresh=0.05)
        xy, conf, d, _ = e.get_points(torch_gpu.from_numpy(pm.reshape(1, h, w)), torch_gpu.from_numpy(g["c%d_map" % i]))[0]
        assert len(conf) == len(uniq)
        col = {int(f): int(j) for f, j in zip(uniq, first)}
        want = np.stack([ref[:, col[int(y) * w + int(x)]] for x, y in xy])
        np.testing.assert_allclose(d, want, rtol=0, atol=2e-6)
        e.close()


def test_homography_adaptation_against_oracle_flow(torch_gpu):
    """fpc_homography_adaptation (python/src/homographies.py:250-324) vs the oracle's restatement of the same flow
    (oracle network, oracle warps / erosion / aggregation) on two small frames, four views, both aggregations.
    The two forwards differ by fp32 noise and a validity mask can flip where a source coordinate lands within fp32
    noise of a pixel boundary, so the comparison tolerates a handful of pixels."""
    from fpc_amd.inference import HomographyConfig, sample_homography, InferenceWrapper, SuperPointSettings  # noqa: F401
    oracle = oracle_mod()
    h, w, n = 64, 96, 2
    sd = synth.make_state_dict(3, dustbin_bias=2.0)
    frames = synth.make_batch(5, n, h, w)
    rng = np.random.default_rng(11)
    hs = np.stack([sample_homography((h, w), HomographyConfig(), rng) for _ in range(4)])
    e = engine(h, w, n)
    e.load_state_dict(sd)
    fwd = lambda f: oracle.forward(f, sd, SPEC)[0]                       # noqa: E731
    for agg, radius in (("sum", 4), ("max", 0)):
        got = e.homography_adaptation(frames, hs, None, radius, agg).cpu().numpy()
        want = oracle.homography_adaptation(frames, fwd, hs, None, radius, agg)
        bad = np.abs(got - want) > 1e-4
        assert bad.mean() < 5e-3, (agg, float(bad.mean()))
        assert got.shape == (n, h, w) and got.max() > 0.01
    # the reference-style entry point (inferencewrapper.py:48-68): one point array per frame, as get_points on the maps
    import torch
    wrap = InferenceWrapper.__new__(InferenceWrapper)
    wrap.name, wrap.settings = "SuperPoint", SuperPointSettings()
    from fpc_amd.inference import SuperPoint as _SP
    wrap.net = _SP(wrap.settings, 0, n)
    wrap.net.load_state_dict(sd)
    cfg = HomographyConfig()
    cfg.num, cfg.valid_border_margin = 4, 4
    pts = wrap.run_with_homography_adaptation(torch.from_numpy(frames), cfg, homographies=hs)
    want = oracle.homography_adaptation(frames, fwd, hs, None, 4, "sum")
    assert len(pts) == n
    for i in range(n):
        oxs, oys, oconf, _ = oracle.get_points(want[i])
        so, sg = set(zip(oxs.tolist(), oys.tolist())), set(zip(pts[i][0].astype(int).tolist(), pts[i][1].astype(int).tolist()))
        assert len(so & sg) >= 0.95 * max(len(so), len(sg)) and pts[i].shape[0] == 3
    # zero views: the plain probability map
    p0 = e.homography_adaptation(frames, np.zeros((0, 8), np.float32), None, 0, "sum").cpu().numpy()
    np.testing.assert_allclose(p0, e.forward(frames)[0].cpu().numpy(), rtol=0, atol=1e-7)
    e.close()


def test_resized_camera_frames_on_device(torch_gpu, golden_dir):
    """fpc_detect_u8_resized: make_query_image (inference.py:72-85) fused with the 8-bit conversion -- bit-exact against
    the oracle (itself held to torch's F.interpolate by fixture F9), on the fixture's frames and on a 720p-like frame
    resized to the engine's VGA; keypoints equal fpc_detect on the converted frames."""
    oracle = oracle_mod()
    g = np.load(os.path.join(golden_dir, "f9_query_image.npz"))
    sd = synth.make_state_dict(3, dustbin_bias=2.0)
    for i in (0, 1, 2, 3):
        th, tw = [int(v) for v in g["c%d_hw" % i]]
        if th % 16 or tw % 16:
            continue
        e = engine(th, tw, 2)
        e.load_state_dict(sd)
        fr = np.stack([g["c%d_frame" % i], g["c%d_frame" % i][::-1].copy()])
        res = e.detect_u8_resized(fr, "bgr_hwc")
        want = oracle.resize_crop_u8(fr, th, tw, swap_rb=True)
        np.testing.assert_array_equal(e.u8_staging(2).cpu().numpy(), want)
        np.testing.assert_allclose(want[0], g["c%d_out" % i], rtol=0, atol=1e-5)
        ref = e.detect(want)
        for a, b in zip(res, ref):
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
        e.close()
    rng = np.random.Generator(np.random.PCG64(4))
    cam = rng.integers(0, 256, size=(1, 360, 640, 3), dtype=np.uint8)        # 16:9 camera frame -> 4:3 network input
    e = engine(240, 320, 1)
    e.load_state_dict(sd)
    e.detect_u8_resized(cam, "rgb_hwc")
    np.testing.assert_array_equal(e.u8_staging(1).cpu().numpy(), oracle.resize_crop_u8(cam, 240, 320, swap_rb=False))
    e.close()


def test_random_geometries_and_settings(torch_gpu):
    """Seeded sweep over frame sizes that do not divide into the kernels' tiles (8x16, 6x20, 3x20 pixels; 1/8 and
    1/16 resolution maps down to 2 x 3 cells), batch sizes, arithmetic modes and post-processing settings: dense maps
    within 1e-4 of the oracle, post-processing exact on the engine's own maps."""
    oracle = oracle_mod()
    rng = np.random.Generator(np.random.PCG64(2718))
    sizes = [(32, 48), (48, 80), (80, 48), (112, 176), (96, 320), (176, 64), (144, 208), (64, 336)]
    for case in range(24):
        h, w = sizes[case % len(sizes)]
        n = int(rng.integers(1, 6))
        dtype = ["f32", "f32_split", "f32_split_f16", "f32"][case % 4]
        de = bool(case % 4 != 3)
        nms, border = int(rng.integers(0, 7)), int(rng.integers(0, 7))
        thr = float(rng.choice([0.005, 0.015, 0.05]))
        sd = synth.make_state_dict(100 + case, dustbin_bias=float(rng.choice([2.0, 4.0])))
        frames = synth.make_batch(1000 + 10 * case, n, h, w)
        # every second fp32 case on the batch plan's kernels (F(4x4,3x3) tiles of 16 x 16 / 8 x 32 pixels on these odd sizes)
        plan = dict(plan_flags=["no_latency_tiles"]) if dtype == "f32" and case % 2 == 0 else {}
        e = engine(h, w, n, dtype=dtype, descriptor_enabled=de, nms_dist=nms, border_remove=border, conf_thresh=thr, **plan)
        e.load_state_dict(sd if de else {k: v for k, v in sd.items() if not k.startswith("descriptor.")})
        prob, desc, logits = e.forward(frames)
        res = e.detect(frames)
        i = int(rng.integers(0, n))
        o_prob, o_desc, o_logits = oracle.forward(frames[i:i + 1], sd, SPEC, descriptor_enabled=de)
        tag = "case %d: %dx%d n=%d %s de=%d" % (case, h, w, n, dtype, de)
        assert np.max(np.abs(logits[i].cpu().numpy() - o_logits[0])) < ATOL, tag
        assert np.max(np.abs(prob[i].cpu().numpy() - o_prob[0])) < ATOL, tag
        if de:
            assert np.max(np.abs(desc[i].cpu().numpy() - o_desc[0])) < ATOL, tag
        pm = prob[i].cpu().numpy()
        oxs, oys, oconf, oncand = oracle.get_points(pm, conf_thresh=thr, nms_dist=nms, border_remove=border)
        xy, conf, d, ncand = res[i]
        assert ncand == oncand, tag
        np.testing.assert_array_equal(xy[:, 0], oxs, err_msg=tag)
        np.testing.assert_array_equal(xy[:, 1], oys, err_msg=tag)
        np.testing.assert_array_equal(conf, oconf, err_msg=tag)
        if de and len(oxs):
            od = oracle.get_descriptors(desc[i].cpu().numpy(), oxs, oys, h, w)
            finite = np.isfinite(od).all(axis=1)          # an all-zero sampled vector normalises to NaN, as in the reference
            np.testing.assert_allclose(d[finite], od[finite], rtol=0, atol=2e-6, err_msg=tag)
        e.close()


def test_random_geometries_other_networks_and_modes(torch_gpu):
    """The same sweep for the C++ network (all three fp32-class modes, against oracle_vgg_forward) and, for bounds
    safety, the bf16 mode (agreement with the fp32 path at bf16 accuracy)."""
    oracle = oracle_mod()
    vspec = arch.vgg_state_dict_spec()
    for case, (h, w) in enumerate([(40, 56), (72, 136), (104, 48), (24, 200)]):
        sd = synth.make_vgg_state_dict(40 + case, 3.0)
        fr = _gray(600 + case, 2, h, w)
        o_prob, o_desc, o_logits = oracle.vgg_forward(fr[1:2], sd, vspec)
        for dtype in ("f32", "f32_split", "f32_split_f16"):
            e = engine(h, w, 2, in_channels=1, arch="vgg", dtype=dtype)
            e.load_state_dict(sd)
            prob, desc, logits = e.forward(fr)
            tag = "vgg %dx%d %s" % (h, w, dtype)
            assert np.max(np.abs(logits[1].cpu().numpy() - o_logits[0])) < ATOL, tag
            assert np.max(np.abs(desc[1].cpu().numpy() - o_desc[0])) < ATOL, tag
            res = e.detect(fr)
            _check_frame_against_oracle_postproc(oracle, prob[1].cpu().numpy(), desc[1].cpu().numpy(), res[1], h, w)
            e.close()
    for case, (h, w) in enumerate([(48, 80), (112, 176), (176, 64)]):
        sd = synth.make_state_dict(70 + case, dustbin_bias=2.0)
        fr = synth.make_batch(800 + case, 3, h, w)
        a = engine(h, w, 3)
        a.load_state_dict(sd)
        b = engine(h, w, 3, dtype="bf16")
        b.load_state_dict(sd)
        la, lb = a.forward(fr)[2], b.forward(fr)[2]
        assert float((la - lb).abs().max()) < BF16_LOGIT_MAX, (h, w)
        assert len(b.detect(fr)) == 3
        b.forward(fr)     # (the taps below include the logits, which only a forward writes in this mode)
        # every layer against the bf16-emulating oracle at these sizes too (partial tiles on every edge of the persistent
        # kernels' tile walks; the last call ran the same frames)
        _check_bf16_layers(oracle, b, fr, sd, [0, 2], "%dx%d" % (h, w))
        a.close()
        b.close()
    # bf16 with a gray plane (stem_pool_bf16_kernel<1>): what the three-channel engine gives on the replicated plane, at bf16
    # accuracy (the gray stem's weights are the sum over the input channels, rounded once)
    h, w = 112, 208
    sd = synth.make_state_dict(77, dustbin_bias=4.0)
    rgb = np.stack([synth.make_frame(900 + i, h, w, gray=True).transpose(2, 0, 1) for i in range(2)])
    b3 = engine(h, w, 2, dtype="bf16")
    b1 = engine(h, w, 2, dtype="bf16", in_channels=1)
    b3.load_state_dict(sd)
    b1.load_state_dict(sd)
    l3, l1 = b3.forward(rgb)[2], b1.forward(np.ascontiguousarray(rgb[:, :1]))[2]
    assert float((l3 - l1).abs().max()) < BF16_LOGIT_MAX and float((l3 - l1).float().pow(2).mean().sqrt()) < BF16_LOGIT_RMS
    ov, cos = _overlap_and_cosine(b3.detect(rgb)[0], b1.detect(np.ascontiguousarray(rgb[:, :1]))[0])
    assert ov >= BF16_KP_OVERLAP and cos >= BF16_DESC_COS
    b3.close()
    b1.close()
    # conf_thresh = 0 in bf16: every pixel is a candidate -- the fused exp-softmax epilogue of the detector's last block
    # fills its per-tile candidate list to capacity; same keypoints, bit for bit, as the separate softmax launch
    h, w = 48, 176
    sd = synth.make_state_dict(78, dustbin_bias=1.0)
    fr = synth.make_batch(950, 2, h, w)
    f1 = engine(h, w, 2, dtype="bf16", conf_thresh=0.0)
    f0 = engine(h, w, 2, dtype="bf16", conf_thresh=0.0, plan_flags=["no_fused_softmax"])
    f1.load_state_dict(sd)
    f0.load_state_dict(sd)
    for ra, rb in zip(f1.detect(fr), f0.detect(fr)):
        assert ra[3] == rb[3] == h * w
        for x, y in zip(ra[:3], rb[:3]):
            np.testing.assert_array_equal(x, y)
    f1.close()
    f0.close()
