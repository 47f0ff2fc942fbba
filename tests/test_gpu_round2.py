"""Round-2 parity tests (pytest -m gpu): BASELINE.json configs[0] at its stated size through the C++ entry point,
the real two-process start-up exchange on one GPU, the C-ABI additions (fpc_sample_descriptors,
fpc_broadcast_weights, fpc_config plan fields), and the edge cases the round-1 advisor named (conf_thresh == 0 with
exact zeros in the map, max_keypoints below the kept count, a tampered packed blob)."""
import ctypes
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import fpc_amd  # noqa: F401
from fpc_amd import arch, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC = arch.state_dict_spec()
ATOL = 1e-4


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


def test_config0_gray_vga_frame_through_the_cpp_entry_point(torch_gpu, golden_dir, tmp_path):
    """BASELINE.json configs[0]: ONE 640x480 grayscale frame, a checkpoint FILE in the reference's layout, the C++
    entry point (superpoint::SuperPoint(file, false).ProcessFrame(gray), cpp/src/superpoint.h:14-18) -- against the
    REFERENCE's outputs for that frame (fixture f5_e2e_gray_vga: the Python network on the plane replicated x3,
    dataset_utils.py:19-20): identical keypoint set, confidences and descriptors within 1e-4."""
    torch = torch_gpu
    g = np.load(os.path.join(golden_dir, "f5_e2e_gray_vga.npz"))
    h, w = int(g["h"]), int(g["w"])
    assert (h, w) == (480, 640)
    demo = os.path.join(ROOT, "feature-point-cnn_amd", "lib", "fpc_demo")
    assert os.path.exists(demo), "run __graft_entry__.build()"
    sd = synth.make_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    ck = str(tmp_path / "super_point_0.pt")
    torch.save({"epoch": 0, "model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in sd.items()},
                "optimizer_state_dict": {}, "scaler_state_dict": {}}, ck)
    gray = synth.make_frame(int(g["seed_frame"]), h, w, gray=True)
    gray[..., 0].tofile(str(tmp_path / "frame.f32"))            # CV_32FC1, H x W, [0,1]: cpp/src/camera.cc:16-18
    out = str(tmp_path / "pts.txt")
    msg = subprocess.check_output([demo, ck, str(tmp_path / "frame.f32"), str(h), str(w), out]).decode()
    got = np.loadtxt(out, ndmin=2)
    gx, gy, gc = g["points_x"].astype(np.int64), g["points_y"].astype(np.int64), g["points_conf"]
    assert msg.startswith("%d feature points" % len(gc))
    mine = got[:, 1].astype(np.int64) * w + got[:, 0].astype(np.int64)
    np.testing.assert_array_equal(np.sort(mine), np.sort(gy * w + gx))       # identical indices after NMS
    assert np.all(np.diff(got[:, 2]) <= 0)
    np.testing.assert_allclose(got[:, 2], gc, rtol=0, atol=ATOL)            # same rank -> same confidence
    # descriptors: the demo prints components 0..2 and 127 of every point; the fixture holds every 16th point
    order = {int(i): k for k, i in enumerate((gy * w + gx).tolist())}
    ref_rank = np.array([order[int(i)] for i in mine])
    inv = np.empty_like(ref_rank)
    inv[ref_rank] = np.arange(len(ref_rank))
    rows = inv[g["desc_subset_idx"]]
    np.testing.assert_allclose(got[rows, 3:6], g["desc_subset"][:, 0:3], rtol=0, atol=ATOL)
    np.testing.assert_allclose(got[rows, 6], g["desc_subset"][:, 127], rtol=0, atol=ATOL)
    # the same frame through the Python host with a gray context gives the demo's numbers exactly
    e = engine(h, w, in_channels=1)
    e.load_state_dict(sd)
    xy, conf, d, ncand = e.detect(np.ascontiguousarray(gray[..., :1].transpose(2, 0, 1)[None]))[0]
    assert ncand == int(g["n_candidates"])
    np.testing.assert_array_equal(got[:, 0].astype(np.int32), xy[:, 0])
    np.testing.assert_array_equal(got[:, 1].astype(np.int32), xy[:, 1])
    # and the dense maps against the reference's probes
    prob, desc, logits = e.forward(np.ascontiguousarray(gray[..., :1].transpose(2, 0, 1)[None]))
    np.testing.assert_allclose(logits.cpu().numpy().ravel()[::7], g["logits_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(desc.cpu().numpy().ravel()[::11], g["desc_map_probe"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(prob.cpu().numpy().ravel()[::13], g["prob_probe"], rtol=0, atol=ATOL)
    e.close()
    # settings() changed between two frames of the SAME size takes effect at the next frame (round 3 applied it only when
    # the frame size changed): the demo's second pass with nms_dist = 8 reports what a context built with 8 reports
    msg2 = subprocess.check_output([demo, ck, str(tmp_path / "frame.f32"), str(h), str(w), out, "8"]).decode().splitlines()
    e8 = engine(h, w, in_channels=1, nms_dist=8)
    e8.load_state_dict(sd)
    k8 = len(e8.detect(np.ascontiguousarray(gray[..., :1].transpose(2, 0, 1)[None]))[0][1])
    e8.close()
    assert msg2[0].startswith("%d feature points" % len(gc)) and msg2[1] == "%d feature points with nms_dist 8" % k8
    assert k8 < len(gc)


def test_get_descriptors_on_its_own_against_reference_fixtures(torch_gpu, golden_dir):
    """fpc_sample_descriptors / inference.get_descriptors (netutils.py:103-121) at caller-provided points, directly
    against the reference's outputs F4b (D = 128; wide, tall and HD-like maps; border rows and columns) -- no planted
    probability map in between -- plus fractional coordinates against the oracle."""
    torch = torch_gpu
    from fpc_amd.inference import SuperPointSettings, get_descriptors
    g = np.load(os.path.join(golden_dir, "f4b_get_descriptors_random.npz"))
    oracle = oracle_mod()
    for i in range(5):
        h, w = (int(v) for v in g["c%d_hw" % i])
        if h % 8 or w % 8 or h < 16 or w < 16:
            continue
        dm, pts, want = g["c%d_map" % i], g["c%d_pts" % i].astype(np.float64), g["c%d_out" % i]
        e = engine(h, w, descriptor_enabled=False)          # sampling needs no descriptor head and no weights
        got = e.sample_descriptors(torch.from_numpy(dm), pts.T)
        np.testing.assert_allclose(got.T, want, rtol=0, atol=2e-6)
        # the reference-shaped function: points float64 [3,K], map [1,D,Hc,Wc] -> [D,K]
        p3 = np.vstack([pts, np.zeros((1, pts.shape[1]))])
        out = get_descriptors(p3, torch.from_numpy(dm).cuda(), h, w, SuperPointSettings(), engine=e)
        assert out.shape == want.shape
        np.testing.assert_allclose(out, want, rtol=0, atol=2e-6)
        assert get_descriptors(np.zeros((3, 0)), torch.from_numpy(dm), h, w, SuperPointSettings(), engine=e).shape == (128, 0)
        e.close()
    # fractional and out-of-frame coordinates (zero padding): the reference's arithmetic in the oracle
    h, w = 64, 96
    rng = np.random.Generator(np.random.PCG64(3))
    dm = rng.normal(0, 1, (1, 128, h // 8, w // 8)).astype(np.float32)
    xy = np.stack([rng.uniform(-3, w + 3, 200), rng.uniform(-3, h + 3, 200)], 1)
    e = engine(h, w, descriptor_enabled=False)
    got = e.sample_descriptors(torch.from_numpy(dm), xy)
    want = oracle.get_descriptors_at(dm[0], xy[:, 0], xy[:, 1], h, w)
    ok = np.isfinite(want).all(1)
    assert ok.sum() > 150
    np.testing.assert_allclose(got[ok], want[ok], rtol=0, atol=3e-6)
    e.close()


def test_threshold_zero_and_exact_zeros_in_the_map(torch_gpu):
    """`prob >= thresh` (netutils.py:59) with thresh == 0 and with a tiny positive thresh on maps that contain exact
    zeros: the NMS state word of a candidate with p == 0.0 must not alias 'empty' (round-1 advisor: the finishing
    kernel would then wait forever on a cell that never changes).  conf_thresh < 0 / NaN is refused at fpc_create."""
    torch = torch_gpu
    from fpc_amd import _lib
    oracle = oracle_mod()
    h, w = 32, 48
    rng = np.random.Generator(np.random.PCG64(17))
    pm = rng.uniform(0.0, 1.0, (1, h, w)).astype(np.float32)
    pm[0, rng.integers(0, h, 300), rng.integers(0, w, 300)] = 0.0        # exact zeros, some adjacent to each other
    pm[0, 10:14, 20:30] = 0.0                                            # a block of them
    pm[0, 5, 5] = -0.0
    for thresh in (0.0, 1e-30, 0.5):
        e = engine(h, w, conf_thresh=thresh, descriptor_enabled=False)
        xy, conf, _, ncand = e.get_points(torch.from_numpy(pm))[0]
        oxs, oys, oconf, oncand = oracle.get_points(pm[0], conf_thresh=thresh)
        assert ncand == oncand and (thresh > 0 or ncand == h * w)
        np.testing.assert_array_equal(xy[:, 0], oxs)
        np.testing.assert_array_equal(xy[:, 1], oys)
        np.testing.assert_array_equal(conf, oconf)
        e.close()
    # an all-zero map with thresh 0: every pixel is a candidate with the same confidence; ties go to the lower index
    e = engine(h, w, conf_thresh=0.0, descriptor_enabled=False, border_remove=0)
    z = np.zeros((1, h, w), np.float32)
    xy, conf, _, ncand = e.get_points(torch.from_numpy(z))[0]
    oxs, oys, oconf, oncand = oracle.get_points(z[0], conf_thresh=0.0, border_remove=0)
    assert ncand == h * w == oncand and len(conf) == len(oconf) and not conf.any()
    np.testing.assert_array_equal(xy[:, 0], oxs)
    np.testing.assert_array_equal(xy[:, 1], oys)
    e.close()
    for bad in (-0.5, float("nan"), float("inf")):
        with pytest.raises(_lib.FpcError) as ei:
            engine(h, w, conf_thresh=bad)
        assert ei.value.code == -1
    # a batch whose tensors 32-bit byte offsets cannot address is refused at fpc_create, before any allocation
    with pytest.raises(_lib.FpcError) as ei:
        engine(960, 1280, b=256)
    assert ei.value.code == -1


def test_get_points_across_the_slices_of_the_candidate_list(torch_gpu):
    """The survivors of a frame are sorted in slices of its candidate list by several workgroups and merged by rank
    (nms_chunk_sort_kernel / nms_merge_kernel): every regime of that path against the oracle, bit for bit, on VGA maps --
    a typical density (4 slices), just above 8 slices' worth (longer slices), every pixel a candidate (38 400 per
    slice), and nms_dist = 0 on a dense map (more survivors in a slice than LDS holds: the one-workgroup kernel redoes
    the frame); the same frames through the one-workgroup plan must agree too."""
    torch = torch_gpu
    oracle = oracle_mod()
    h, w = 480, 640
    rng = np.random.Generator(np.random.PCG64(23))
    u = rng.uniform(0.0, 1.0, (h, w)).astype(np.float32)
    cases = [(np.where(u > 0.978, u, 0.0), 0.015, 4, 4),          # ~6 800 candidates
             (np.where(u > 0.94, u, 0.0), 0.015, 4, 4),           # ~18 400 > 8 x 2048
             (np.maximum(u, 0.02), 0.015, 4, 4),                  # all 307 200
             (np.where(u > 0.9, u, 0.0), 0.015, 0, 2),            # ~30 700 candidates, all of them survive
             (np.where(u > 0.978, (u * 8).astype(np.int32) / 8.0, 0.0), 0.015, 4, 0)]   # ties everywhere
    for i, (m, thr, r, bw) in enumerate(cases):
        m = np.ascontiguousarray(m, dtype=np.float32)
        oxs, oys, oconf, oncand = oracle.get_points(m, thr, r, bw)
        for flags in ([], ["nms_one_workgroup"]):
            e = engine(h, w, descriptor_enabled=False, nms_dist=r, border_remove=bw, conf_thresh=thr, plan_flags=flags)
            xy, conf, _, ncand = e.get_points(torch.from_numpy(m[None]))[0]
            assert ncand == oncand, (i, flags)
            np.testing.assert_array_equal(xy[:, 0], oxs, err_msg=str((i, flags)))
            np.testing.assert_array_equal(xy[:, 1], oys, err_msg=str((i, flags)))
            np.testing.assert_array_equal(conf, oconf, err_msg=str((i, flags)))
            e.close()


def test_max_keypoints_keeps_the_most_confident_points(torch_gpu):
    """fpc_config.max_keypoints below the kept count: the frame stays readable and holds the `cap` best points
    (round 1 returned FPC_E_CAPACITY for such a frame)."""
    h, w = 96, 128
    sd = synth.make_state_dict(31, dustbin_bias=5.0)
    frames = synth.make_batch(40, 2, h, w)
    full = engine(h, w, 2)
    full.load_state_dict(sd)
    ref = full.detect(frames)
    assert min(len(r[1]) for r in ref) > 40
    cap = 25
    e = engine(h, w, 2, max_keypoints=cap)
    e.load_state_dict(sd)
    assert e.capacity == cap
    for (xy, conf, d, nc), (rxy, rconf, rd, rnc) in zip(e.detect(frames), ref):
        assert len(conf) == cap and nc == rnc
        np.testing.assert_array_equal(xy, rxy[:cap])
        np.testing.assert_array_equal(conf, rconf[:cap])
        np.testing.assert_array_equal(d, rd[:cap])
    full.close()
    e.close()


def test_plan_fields_of_fpc_config(torch_gpu, golden_dir):
    """The launch-plan knobs as fpc_config fields (num_streams, plan_flags, nms_round_launches, min_sub_batch): every
    alternative plan gives the default plan's keypoints (sets; bit-equal where the arithmetic is the same) and the
    dense maps within 1e-4."""
    from fpc_amd import _lib
    h, w, n = 96, 128, 6
    sd = synth.make_state_dict(31, dustbin_bias=5.0)
    frames = synth.make_batch(40, n, h, w)
    base = engine(h, w, n)
    base.load_state_dict(sd)
    ref = base.detect(frames)
    _, _, lref = base.forward(frames)
    plans = [dict(num_streams=1), dict(num_streams=3, min_sub_batch=2), dict(nms_round_launches=-1),
             dict(nms_round_launches=5), dict(plan_flags=["nms_one_workgroup"]), dict(plan_flags=["no_winograd"]), dict(plan_flags=["no_fused_blocks"]),
             dict(plan_flags=["no_winograd_detector", "no_winograd_layer_in1", "no_xcd_order"]),
             dict(plan_flags=["no_fused_stem_pool", "nms_in_line"]), dict(plan_flags=["split_heads", "no_persistent_grid"]),
             dict(plan_flags=["no_winograd", "layer1_tile_8x16"]), dict(plan_flags=sum(_lib.PLAN_FLAGS.values()) & ~(1 << 6)),
             dict(plan_flags=["heads_in_line"])]       # (round 4: the two heads side by side are this plan's default)
    for kw in plans:
        e = engine(h, w, n, **kw)
        e.load_state_dict(sd)
        got = e.detect(frames)
        _, _, lg = e.forward(frames)
        assert float((lg - lref).abs().max()) < ATOL, kw
        # a call below 2 * min_sub_batch frames takes the latency plan (layer_in.1 as the fused direct block instead of
        # the Winograd launches): min_sub_batch = 2 moves these 6 frames to the other side, i.e. to other arithmetic
        same_arith = kw.get("plan_flags", []) in ([], ["nms_one_workgroup"], ["heads_in_line"]) and not kw.get("min_sub_batch")
        for (xy, conf, d, nc), (rxy, rconf, rd, rnc) in zip(got, ref):
            if same_arith:      # stream / NMS-launch plans do not touch the arithmetic: bit-equal
                assert nc == rnc
                np.testing.assert_array_equal(xy, rxy)
                np.testing.assert_array_equal(conf, rconf)
                np.testing.assert_array_equal(d, rd)
            else:
                a, b = set(map(tuple, xy.tolist())), set(map(tuple, rxy.tolist()))
                assert len(a & b) >= 0.99 * max(len(a), len(b)), kw
        e.close()
    base.close()


def test_packed_blob_carries_a_tag(torch_gpu):
    """The packed weight blob names its own layout: an engine of another dtype / plan refuses it, a flipped tag byte is
    refused, and a context reports the same plan hash iff blobs are exchangeable."""
    from fpc_amd import _lib
    h, w = 32, 48
    sd = synth.make_state_dict(2)
    a = engine(h, w)
    a.load_state_dict(sd)
    blob = a.export_packed()
    assert blob[:4].tobytes() == b"FPCW"
    b = engine(64, 96, 4)                          # geometry and batch do not enter the layout
    assert b.plan_hash() == a.plan_hash() and b.packed_size() == a.packed_size()
    b.import_packed(blob)
    fr = synth.make_batch(9, 1, 64, 96)
    a2 = engine(64, 96)
    a2.load_state_dict(sd)
    for x, y in zip(b.detect(fr)[0][:3], a2.detect(fr)[0][:3]):
        np.testing.assert_array_equal(x, y)
    other = [engine(h, w, dtype="f32_split"), engine(h, w, plan_flags=["no_winograd"]), engine(h, w, descriptor_enabled=False)]
    for o in other:
        assert o.plan_hash() != a.plan_hash()
        if o.packed_size() == blob.size:
            with pytest.raises(_lib.FpcError) as ei:
                o.import_packed(blob)
            assert ei.value.code == -1 and "packed weights" in str(ei.value)
        o.close()
    bad = blob.copy()
    bad[17] ^= 0x40                                 # a byte of the plan hash
    c = engine(h, w)
    with pytest.raises(_lib.FpcError):
        c.import_packed(bad)
    with pytest.raises(_lib.FpcError):              # nothing was written in place: the device blob holds no tag
        c.mark_weights_loaded()
    with pytest.raises(_lib.FpcError) as ei:
        c.forward(np.zeros((1, 3, h, w), np.float32))
    assert ei.value.code == -4
    for x in (a, a2, b, c):
        x.close()


def _rccl():
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return ctypes.CDLL(path if os.path.exists(path) else "librccl.so.1", mode=ctypes.RTLD_GLOBAL)


def test_fpc_broadcast_weights_on_an_rccl_communicator(torch_gpu):
    """fpc_broadcast_weights(ctx, ncclComm_t, root) -- the C-ABI's own start-up exchange -- on a real RCCL communicator.
    This box has ONE GPU and RCCL refuses two ranks on one device, so the communicator has one rank: that still runs
    librccl resolution, ncclCommUserRank, both ncclBroadcast calls on the ctx stream and the tag checks."""
    from fpc_amd import _lib
    rccl = _rccl()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    h, w = 32, 48
    e = engine(h, w)
    with pytest.raises(_lib.FpcError) as ei:        # a root without weights: every rank is told so after the tag
        e.broadcast_weights(comm.value, 0)
    assert ei.value.code == -4
    sd = synth.make_state_dict(2)
    e.load_state_dict(sd)
    before = e.detect(synth.make_batch(9, 1, h, w))[0]
    e.broadcast_weights(comm.value, 0)              # root == this rank: in-place broadcast of the blob onto itself
    after = e.detect(synth.make_batch(9, 1, h, w))[0]
    for x, y in zip(before[:3], after[:3]):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(_lib.FpcError):
        e.broadcast_weights(None, 0)
    e.close()
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)


_RANK_SCRIPT = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
import fpc_amd
from fpc_amd import dist as fdist, synth
from fpc_amd.engine import Engine
rank, world, _ = fdist.init_from_env(backend="gloo")
h, w, n = 64, 96, 6
sd = synth.make_state_dict(3, dustbin_bias=4.0) if rank == 0 else None     # only rank 0 ever sees the checkpoint
eng = Engine(h, w, max_batch=n)
fdist.broadcast_packed_weights(eng, sd)
frames = synth.make_batch(50, n, h, w)
lo, hi = fdist.shard_range(n, world, rank)
mine = eng.detect(frames[lo:hi])                 # this rank's contiguous shard
everything = eng.detect(frames)                  # and the whole batch, to compare the ranks with each other
# failure path 1: the source rank cannot load -> every rank raises, nobody hangs
bad = Engine(h, w)
try:
    fdist.broadcast_packed_weights(bad, {} if rank == 0 else None)
    fail1 = False
except fdist.WeightBroadcastError:
    fail1 = True
# failure path 2: rank 1's engine has another dtype -> every rank raises before the blob is sent
odd = Engine(h, w, dtype="f32_split" if rank == 1 else "f32")
try:
    fdist.broadcast_packed_weights(odd, synth.make_state_dict(3, dustbin_bias=4.0) if rank == 0 else None)
    fail2 = False
except fdist.WeightBroadcastError:
    fail2 = True
fdist.barrier()
with open(os.environ["FPC_TEST_OUT"] + ".%%d" %% rank, "wb") as f:
    pickle.dump(dict(rank=rank, lo=lo, hi=hi, mine=mine, everything=everything, fail1=fail1, fail2=fail2,
                     plan=eng.plan_hash()), f)
dist.destroy_process_group()
'''


def test_two_ranks_with_real_engines_on_one_gpu(torch_gpu, tmp_path):
    """The N > 1 start-up path with the REAL Engine in two processes (gloo; RCCL refuses two ranks on one device):
    rank 0 parses + packs, export_packed -> broadcast -> import_packed, then each rank detects its contiguous frame
    shard.  Rank 1 -- which never saw the checkpoint -- must detect bit-identically to rank 0, shards must tile the
    batch, and both failure paths must raise on both ranks instead of hanging."""
    import pickle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT})
    out = str(tmp_path / "res")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FPC_TEST_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    res = [pickle.load(open(out + ".%d" % r, "rb")) for r in range(2)]
    assert res[0]["plan"] == res[1]["plan"]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 3, 3, 6)
    assert all(r["fail1"] and r["fail2"] for r in res)
    for a, b in zip(res[0]["everything"], res[1]["everything"]):       # rank 1 == rank 0, bit for bit, on every frame
        assert a[3] == b[3]
        for x, y in zip(a[:3], b[:3]):
            np.testing.assert_array_equal(x, y)
    whole = res[0]["everything"]
    for r in res:                                                       # a shard is the same frames of the whole batch
        for k, fr in enumerate(r["mine"]):
            for x, y in zip(fr[:3], whole[r["lo"] + k][:3]):
                np.testing.assert_array_equal(x, y)
    # and against the oracle on rank 1's shard: the blob it received is the right one
    from fpc_amd.engine import Engine
    e = Engine(64, 96, max_batch=6)
    sd = synth.make_state_dict(3, dustbin_bias=4.0)
    e.load_state_dict(sd)
    frames = synth.make_batch(50, 6, 64, 96)
    oracle = oracle_mod()
    prob, desc, _ = e.forward(frames)
    for k in range(3, 6):
        oxs, oys, oconf, _ = oracle.get_points(prob[k].cpu().numpy())
        np.testing.assert_array_equal(res[1]["mine"][k - 3][0][:, 0], oxs)
        np.testing.assert_array_equal(res[1]["mine"][k - 3][0][:, 1], oys)
    e.close()


def test_bench_gpus_2_launches_its_own_ranks_on_one_gpu(torch_gpu):
    """`python bench.py --gpus 2` with no launcher around it (what a SCALE run may do): the parent starts the two
    ranks itself; both share this box's one GPU, so the rendezvous is gloo (RCCL refuses two ranks on one device).
    One JSON line, n_gpus = 2, both ranks reported, exit status 0."""
    import json
    env = dict(os.environ, FPC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-steady-state"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["per_rank"]["frames_per_s"]) == 2
    assert out["config"]["frames_per_step_per_gpu"] == 32


def test_logits_tap_after_a_fused_softmax_detect(torch_gpu):
    """FPC_BF16's fpc_detect takes exp-softmax / threshold in detector.layer.1's epilogue and writes no logits: the
    "det.1" tap must say so (FPC_E_INVALID) rather than hand out stale memory; after fpc_forward, or with the fused
    epilogue switched off, it is the logits again."""
    from fpc_amd import _lib
    h, w = 64, 96
    sd = synth.make_state_dict(3, dustbin_bias=4.0)
    frames = synth.make_batch(5, 2, h, w)
    e = engine(h, w, 2, dtype="bf16")
    e.load_state_dict(sd)
    _, _, logits = e.forward(frames)
    tap = e.activation("det.1", 0, 2)
    np.testing.assert_array_equal(tap.cpu().numpy(), logits.cpu().numpy())
    e.detect(frames)
    with pytest.raises(_lib.FpcError):
        e.activation("det.1", 0, 2)
    assert e.activation("det.0", 0, 2).shape[1] == 65          # the other taps are what the detect call left
    e.forward(frames)
    np.testing.assert_array_equal(e.activation("det.1", 0, 2).cpu().numpy(), logits.cpu().numpy())
    e.close()
    e2 = engine(h, w, 2, dtype="bf16", plan_flags=["no_fused_softmax"])
    e2.load_state_dict(sd)
    e2.detect(frames)
    np.testing.assert_array_equal(e2.activation("det.1", 0, 2).cpu().numpy(), logits.cpu().numpy())
    e2.close()
