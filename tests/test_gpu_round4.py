"""Round-4 parity / safety tests (pytest -m gpu), all through the C-ABI.

* The fault of round 3 by name: a map NARROWER than a wblock36_kernel tile made the store descriptor's masked columns
  land W36_MARKER (2 GiB) behind the tensor -- an out-of-range STORE, which no comparison of the tensor itself can see.
  Contexts created with FPC_PLAN_GUARD_ZONES carry a canary pattern behind every buffer of the workspace and 2 GiB
  behind the last one; fpc_check_guards counts overwritten words.
* The bench's headline plan (the whole 32-frame VGA batch as ONE sub-batch on one stream: 640 tiles on a 216-workgroup
  persistent grid, the 2 x 8 instance at layer1) against the oracle and against the library's default plan.
* A packed blob whose tag names another fragment-layout revision / ABI version is refused.
"""
import os

import numpy as np
import pytest

import fpc_amd  # noqa: F401
from fpc_amd import _lib, arch, synth

pytestmark = pytest.mark.gpu

SPEC = arch.state_dict_spec()
ATOL = 1e-4


def engine(h, w, b=1, **kw):
    from fpc_amd.engine import Engine
    return Engine(h, w, max_batch=b, **kw)


def oracle_mod():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("h,w", [(32, 48), (64, 96), (48, 160)])
def test_maps_narrower_than_a_tile_store_nothing_outside_their_tensors(h, w):
    """wblock36_kernel<NB, 4, 4> (16 x 16 pixels) / <NB, 2, 8> (8 x 32) on maps of 4 x 6 ... 12 x 40 pixels, driven through
    the batch plan (`no_latency_tiles`: calls of a few frames otherwise take the 4 x 16 latency instances), batch 2:
    every canary word behind every workspace buffer -- and in the 2 GiB behind the workspace, where round 3's masked
    stores went -- is unchanged after forward + detect, and the dense maps are the oracle's at 1e-4."""
    sd = synth.make_state_dict(3, dustbin_bias=4.0)
    frames = synth.make_batch(11, 2, h, w)
    e = engine(h, w, 2, plan_flags=["no_latency_tiles", "guard_zones"])
    e.load_state_dict(sd)
    assert e.check_guards() == 0                       # the pattern is in place before anything ran
    prob, desc, logits = e.forward(frames)
    res = e.detect(frames)
    assert e.check_guards() == 0, "a kernel stored outside its tensor"
    o_prob, o_desc, o_logits = oracle_mod().forward(frames, sd, SPEC)
    assert np.max(np.abs(logits.cpu().numpy() - o_logits)) < ATOL
    assert np.max(np.abs(desc.cpu().numpy() - o_desc)) < ATOL
    assert np.max(np.abs(prob.cpu().numpy() - o_prob)) < ATOL
    assert len(res) == 2
    e.close()
    # the facility itself: a context without the flag refuses the call (unless the whole run is under FPC_GUARD_ZONES=1)
    if os.environ.get("FPC_GUARD_ZONES") != "1":
        e2 = engine(h, w, 1)
        with pytest.raises(_lib.FpcError):
            e2.check_guards()
        e2.close()


def test_guard_zones_see_a_planted_store():
    """The checker itself: overwrite four words right behind one buffer and 2 GiB behind the workspace's last buffer
    (through a torch view of the library's memory) -- both are counted."""
    import ctypes
    import torch
    from fpc_amd.engine import _DevArray
    e = engine(32, 48, 1, plan_flags=["guard_zones"])
    res = e._res
    # `count` is one of the carved buffers (int32[B]); its guard zone starts at the next 256-byte boundary
    base = int(res.count)
    view = torch.as_tensor(_DevArray(base + 256, 64), device=e.torch_device)
    view[:16] = 0
    torch.cuda.synchronize()
    assert e.check_guards() == 4
    if os.environ.get("FPC_GUARD_ZONES") == "1":      # (a run under the canary zones checks every context at close: this one
        with pytest.raises(RuntimeError):             # was damaged on purpose)
            e.close()
    else:
        e.close()


def test_headline_plan_one_sub_batch_on_one_stream_matches_oracle_and_default_plan():
    """bench.py's headline runs every 32-frame VGA batch as ONE sub-batch on one stream (num_streams = 1): wblock36 grids
    of 216 workgroups walking 640 tiles in three rounds, the 2 x 8 instance at layer1.  Held to (a) the oracle's own
    forward on two frames at 1e-4, (b) the oracle's post-processing of the device's maps, exactly, (c) the library's
    default plan (two 16-frame sub-batches on two streams): bit-identical keypoints, confidences and descriptors."""
    h, w, n = 480, 640, 32
    sd = synth.make_state_dict(0, dustbin_bias=7.0)
    frames = synth.make_batch(100, n, h, w)
    e1 = engine(h, w, n, num_streams=1)
    e1.load_state_dict(sd)
    prob, desc, logits = e1.forward(frames)
    res1 = e1.detect(frames)
    oracle = oracle_mod()
    for i in (3, 20):
        o_prob, o_desc, o_logits = oracle.forward(frames[i:i + 1], sd, SPEC)
        assert np.max(np.abs(logits[i].cpu().numpy() - o_logits[0])) < ATOL
        assert np.max(np.abs(prob[i].cpu().numpy() - o_prob[0])) < ATOL
        assert np.max(np.abs(desc[i].cpu().numpy() - o_desc[0])) < ATOL
    for i in (0, 17, 31):
        xy, conf, d, ncand = res1[i]
        oxs, oys, oconf, oncand = oracle.get_points(prob[i].cpu().numpy())
        assert ncand == oncand
        np.testing.assert_array_equal(xy[:, 0], oxs)
        np.testing.assert_array_equal(xy[:, 1], oys)
        np.testing.assert_array_equal(conf, oconf)
    e2 = engine(h, w, n)
    e2.import_packed(e1.export_packed())
    res2 = e2.detect(frames)
    for a, b in zip(res1, res2):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
    e1.close()
    e2.close()


def test_blob_of_another_fragment_layout_revision_is_refused():
    """The packed blob's 64-byte tag carries FPC_PACK_LAYOUT_REVISION (word 9) beside magic, ABI version, dtype, arch, plan
    hash and size: a blob that names the previous revision (a file kept from an older build, a rank on another build)
    is refused by fpc_import_packed -- round 3 changed the stem's fragment order under an unchanged tag."""
    e = engine(32, 48, 1)
    e.load_state_dict(synth.make_state_dict(1, dustbin_bias=4.0))
    blob = e.export_packed().copy()
    words = blob.view(np.uint32)
    assert words[0] == 0x57435046 and words[1] == _lib.ABI_VERSION and words[9] == _lib.PACK_LAYOUT_REVISION
    e2 = engine(32, 48, 1)
    e2.import_packed(blob)                              # the untouched blob is accepted
    for word, value in ((9, _lib.PACK_LAYOUT_REVISION - 1), (1, _lib.ABI_VERSION - 1)):
        bad = blob.copy()
        bad.view(np.uint32)[word] = value
        with pytest.raises(_lib.FpcError) as ei:
            e2.import_packed(bad)
        assert ei.value.code == -1                      # FPC_E_INVALID
    e.close()
    e2.close()


def test_homography_adaptation_on_the_batch_plan():
    """fpc_homography_adaptation drives the path once per view; a pseudo-labelling batch (preprocess_coco.py: 16 frames x
    16 passes) takes the BATCH plan's kernels -- F(4x4,3x3) blocks, the detector's 64 + 1-channel instance -- which calls of
    a few frames do not.  Same flow as tests/test_gpu_parity.py::test_homography_adaptation_against_oracle_flow with the
    batch plan forced (`no_latency_tiles`) and the canary zones on: against the oracle's restatement of the flow, and
    bit-identical to the small-call plan's own result on the pixels both consider valid is NOT asked (two kernels, fp32
    noise) -- the oracle bar is."""
    from fpc_amd.inference import HomographyConfig, sample_homography
    oracle = oracle_mod()
    h, w, n = 64, 96, 2
    sd = synth.make_state_dict(3, dustbin_bias=2.0)
    frames = synth.make_batch(5, n, h, w)
    rng = np.random.default_rng(11)
    hs = np.stack([sample_homography((h, w), HomographyConfig(), rng) for _ in range(4)])
    e = engine(h, w, n, plan_flags=["no_latency_tiles", "guard_zones"])
    e.load_state_dict(sd)
    fwd = lambda f: oracle.forward(f, sd, SPEC)[0]                       # noqa: E731
    for agg, radius in (("sum", 4), ("max", 0)):
        got = e.homography_adaptation(frames, hs, None, radius, agg).cpu().numpy()
        want = oracle.homography_adaptation(frames, fwd, hs, None, radius, agg)
        bad = np.abs(got - want) > 1e-4
        assert bad.mean() < 5e-3, (agg, float(bad.mean()))
    assert e.check_guards() == 0
    e.close()


def test_cpp_network_on_the_batch_plan_against_the_reference_binary():
    """The C++ frontend's network (arch = "vgg") on the BATCH plan's kernels -- round 4: its 3x3 layers on the conv-only form
    of wblock36_kernel (F(4x4,3x3); 256 outputs in parts; launches given pointers to their first frame) -- against fixture
    F7, the outputs of the reference's own cpp/src/model.cc (oracle/_ref): dense maps at 1e-4, under the canary zones; and
    against the oracle's restatement over the whole maps."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "f7_vgg_qvga.npz"))
    sd = synth.make_vgg_state_dict(int(g["seed_weights"]), float(g["dustbin_bias"]))
    h, w = int(g["h"]), int(g["w"])
    frame = synth.make_batch(int(g["seed_frame"]), 1, h, w, gray=True)[:, :1]
    frames = np.ascontiguousarray(np.repeat(frame, 3, axis=0))
    e = engine(h, w, 3, in_channels=1, arch="vgg", plan_flags=["no_latency_tiles", "guard_zones"])
    e.load_state_dict(sd)
    prob, desc, logits = e.forward(frames)
    for i in range(3):
        np.testing.assert_allclose(logits[i].cpu().numpy().ravel()[::7], g["logits_probe"], rtol=0, atol=ATOL)
        np.testing.assert_allclose(desc[i].cpu().numpy().ravel()[::11], g["desc_probe"], rtol=0, atol=ATOL)
    o_prob, o_desc, o_logits = oracle_mod().vgg_forward(frames[:1], sd, arch.vgg_state_dict_spec())
    assert np.max(np.abs(logits[2].cpu().numpy() - o_logits[0])) < ATOL
    assert np.max(np.abs(desc[2].cpu().numpy() - o_desc[0])) < ATOL
    assert e.check_guards() == 0
    e.close()


def test_upload_stream_runs_beside_the_context_and_orders_by_events():
    """fpc_upload_stream (round 4): a stream of the ctx's for the CALLER's uploads, probed -- like the ctx's own streams in
    fpc_create -- to share no hardware queue with them.  A batch uploaded on it, an event, fpc_detect on the main stream
    behind that event: the same keypoints, bit for bit, as the plain call; the handle is stable across calls and differs
    from the main stream's; two contexts get different ones."""
    import torch
    h, w, n = 64, 96, 2
    sd = synth.make_state_dict(5, dustbin_bias=4.0)
    frames = synth.make_batch(21, n, h, w)
    e = engine(h, w, n)
    e.load_state_dict(sd)
    want = e.detect(frames)
    up, main = e.upload_stream(), e.torch_stream()
    assert up.cuda_stream != 0 and up.cuda_stream != main.cuda_stream
    assert e.upload_stream().cuda_stream == up.cuda_stream
    host = torch.from_numpy(np.array(frames))      # (pageable: the copy below is staged by the runtime, still on `up`)
    dev = torch.empty_like(host, device=e.torch_device)
    done = torch.cuda.Event()
    with torch.cuda.stream(up):
        dev.copy_(host, non_blocking=True)
        done.record(up)
    main.wait_event(done)
    e.detect_async(dev, n)
    got = e.fetch(n)
    for a, b in zip(want, got):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
    e2 = engine(h, w, n)
    assert e2.upload_stream().cuda_stream not in (up.cuda_stream, main.cuda_stream)
    torch.cuda.synchronize()
    del dev          # (before the contexts go: torch's allocator records an event on every stream a freed block was used on)
    e2.close()
    e.close()


def test_sampled_timing_records_every_nth_call():
    """fpc_set_timing(ctx, n > 1): only every n-th fpc_detect call (the first included) carries the HIP events -- what
    bench.py's timed region uses since the event records on every call measured 0.7 % of the frame rate.  Eight calls
    with n = 4 leave exactly twice one call's records, with the same kernels and plausible durations."""
    h, w, n = 64, 96, 2
    e = engine(h, w, n, num_streams=1)
    e.load_state_dict(synth.make_state_dict(5, dustbin_bias=4.0))
    import torch
    frames = torch.from_numpy(synth.make_batch(21, n, h, w)).to(e.torch_device)
    e.set_timing(True)
    e.detect_async(frames, n)
    e.sync()
    one = e.timings()
    e.set_timing(4)
    for _ in range(8):
        e.detect_async(frames, n)
    e.sync()
    got = e.timings()
    assert len(one) > 10 and len(got) == 2 * len(one)
    assert [t[1] for t in got[:len(one)]] == [t[1] for t in one]
    assert all(0.0 < t[2] < 50.0 for t in got)
    e.set_timing(False)
    e.detect_async(frames, n)
    e.sync()
    assert e.timings() == []
    e.close()
