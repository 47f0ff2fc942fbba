"""The N>1 path on CPU: world_size 2 over gloo.  Covers what bench.py does between ranks --
rendezvous from the environment, contiguous frame shards, the start-up broadcast of the packed
weight blob from rank 0, MAX-over-ranks timing -- with a stand-in for the GPU engine (the
collective logic is backend-independent; nccl == RCCL replaces gloo on the GPU box)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

import fpc_amd  # noqa: F401
from fpc_amd import dist as fdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEngine:
    """Holds a 'packed blob' exactly like Engine does, minus the GPU."""

    def __init__(self, n=4099, plan=0x1234567890abcdef):
        self.blob = np.zeros(n, np.uint8)
        self.loaded = False
        self.plan = plan

    def plan_hash(self):
        return self.plan

    def load_state_dict(self, sd):
        if "blob" not in sd:
            raise KeyError("checkpoint entry: encoder.conv1.weight")
        self.blob[:] = np.frombuffer(sd["blob"], np.uint8)
        self.loaded = True

    def packed_size(self):
        return self.blob.size

    def export_packed(self):
        return self.blob.copy()

    def import_packed(self, buf):
        self.blob[:] = buf
        self.loaded = True


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    eng = FakeEngine()
    payload = np.random.Generator(np.random.PCG64(7)).integers(0, 256, eng.packed_size(), dtype=np.uint8)
    sd = {"blob": payload.tobytes()} if rank == 0 else None
    fdist.broadcast_packed_weights(eng, sd)
    ok = eng.loaded and np.array_equal(eng.blob, payload)
    lo, hi = fdist.shard_range(37, world, rank)
    tmax = fdist.max_over_ranks(1.0 + rank)
    tsum = fdist.sum_over_ranks(hi - lo)
    fdist.barrier()
    # the source rank fails to load: EVERY rank raises instead of waiting in the blob broadcast
    bad_src = FakeEngine()
    try:
        fdist.broadcast_packed_weights(bad_src, {} if rank == 0 else None)
        src_fail = False
    except fdist.WeightBroadcastError as e:
        src_fail = ("could not load" in str(e)) and not bad_src.loaded
    # one rank's engine would lay the blob out differently (other dtype / plan knobs): every rank raises, none imports
    odd = FakeEngine(plan=0x1111 if rank == 1 else 0x1234567890abcdef)
    try:
        fdist.broadcast_packed_weights(odd, sd)
        plan_fail = False
    except fdist.WeightBroadcastError as e:
        plan_fail = "plan" in str(e) and (rank == 0 or not odd.loaded)
    small = FakeEngine(n=4099 if rank == 0 else 2048)
    try:
        fdist.broadcast_packed_weights(small, sd)
        size_fail = False
    except fdist.WeightBroadcastError:
        size_fail = rank == 0 or not small.loaded
    fdist.barrier()   # the group is still usable after the refused exchanges
    q.put((rank, ok, lo, hi, tmax, tsum, src_fail, plan_fail, size_fail))
    torch.distributed.destroy_process_group()


def test_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "rank did not receive rank 0's blob"
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 19, 19, 37)
    assert res[0][4] == 2.0 and res[1][4] == 2.0        # MAX over ranks
    assert res[0][5] == 37.0
    assert all(r[6] for r in res), "a failing source rank must raise on every rank"
    assert all(r[7] for r in res), "a plan mismatch on one rank must raise on every rank"
    assert all(r[8] for r in res), "a size mismatch on one rank must raise on every rank"


def test_shard_ranges_cover_every_frame_once():
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert fdist.shard_range(256, 8, 3) == (96, 128)    # configs[2]: 32 frames per GPU


def test_single_process_needs_no_group():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert fdist.init_from_env() == (0, 1, 0)
    assert fdist.max_over_ranks(3.5) == 3.5
    eng = FakeEngine()
    fdist.broadcast_packed_weights(eng, {"blob": bytes(eng.packed_size())})
    assert eng.loaded


def _bench(args, **env):
    import subprocess
    e = dict(os.environ, **env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` without torch.distributed.run around it: the parent starts two ranks, they
    rendezvous (gloo here), rank 0 prints ONE line and the parent exits 0."""
    import json
    r = _bench(["--gpus", "2"], FPC_BENCH_RENDEZVOUS_ONLY="1", FPC_DIST_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    threads = out.pop("host_threads_per_rank")
    assert out.pop("frame_seed0") == [100, 132] and out.pop("shards") == [[0, 32], [32, 64]] and out.pop("host_threads") == [threads] * 2
    assert out == {"rendezvous": "ok", "world": 2, "ranks": [0, 1], "backend": "gloo"}
    assert 1 <= threads <= max(1, len(os.sched_getaffinity(0)) // 2)      # each rank binds cores // N host threads


def test_bench_timing_reduction_with_four_ranks():
    """The N > 1 bookkeeping of bench.py on 4 gloo ranks that pretend to have run their K steps at different speeds
    (FPC_BENCH_FAKE_STEP_MS): `value` = frames of ALL ranks / the SLOWEST rank's time (max over ranks, the contract's
    rule), `per_rank` lists every rank's own rate in rank order.  The reduction is bench.aggregate_over_ranks, the
    function the real run calls -- so the first 8-GPU lease cannot fail in the bookkeeping."""
    import json
    r = _bench(["--gpus", "4", "--steps", "20"], FPC_BENCH_RENDEZVOUS_ONLY="1", FPC_DIST_BACKEND="gloo",
               FPC_BENCH_FAKE_STEP_MS="3.0,3.2,4.0,2.5")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["world"] == 4 and out["ranks"] == [0, 1, 2, 3] and out["n_gpus"] == 4 and out["steps"] == 20
    assert abs(out["ms_per_step"] - 4.0) < 1e-6                           # the slowest rank
    assert abs(out["value"] - 4 * 32 * 20 / (4.0e-3 * 20)) < 0.5           # whole-job frames / that time = 32 000 frames/s
    want = [32 / 3.0e-3, 32 / 3.2e-3, 32 / 4.0e-3, 32 / 2.5e-3]
    assert all(abs(a - b) < 0.5 for a, b in zip(out["per_rank"]["frames_per_s"], want))
    assert out["per_rank"]["weights_start_up_ms"] == [10.0, 11.0, 12.0, 13.0]


def test_bench_gpus_8_rehearsed_over_gloo():
    """`python bench.py --gpus 8` as the driver's first 8-GPU lease will start it, rehearsed with 8 gloo ranks on the CPU:
    the parent starts its ranks, they rendezvous, and every piece of per-rank bookkeeping of configs[2] (256 VGA frames
    sharded over 8 GPUs) comes out as SURVEY 8(d) config 3 states it -- frame seeds 100 + 32 * rank (100 .. 355 over the
    job), contiguous shards of 32 frames with no gap or overlap, cores // 8 host threads per rank -- and
    aggregate_over_ranks (the function the real run calls) prices the job on the slowest rank.  No scaling curve exists:
    this is bookkeeping, not a measurement."""
    import json
    ms = "2.8,2.9,2.85,3.1,2.8,2.8,2.95,2.8"
    r = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5"], FPC_BENCH_RENDEZVOUS_ONLY="1", FPC_DIST_BACKEND="gloo",
               FPC_BENCH_FAKE_STEP_MS=ms)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                                  # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["world"] == 8 and out["ranks"] == list(range(8)) and out["n_gpus"] == 8 and out["backend"] == "gloo"
    assert out["frame_seed0"] == [100 + 32 * r_ for r_ in range(8)] and out["frame_seed0"][-1] + 31 == 355
    assert out["shards"] == [[32 * r_, 32 * r_ + 32] for r_ in range(8)]
    cores = len(os.sched_getaffinity(0))
    assert out["host_threads"] == [out["host_threads_per_rank"]] * 8 and 1 <= out["host_threads_per_rank"] <= max(1, cores // 8)
    assert abs(out["ms_per_step"] - 3.1) < 1e-6
    assert abs(out["value"] - 8 * 32 * 20 / (3.1e-3 * 20)) < 0.5            # 82 580 frames/s for the whole job
    want = [32 / (float(v) * 1e-3) for v in ms.split(",")]
    assert all(abs(a - b) < 0.5 for a, b in zip(out["per_rank"]["frames_per_s"], want))
    # ragged shards (a caller's batch that does not divide): remainders to the low ranks, still contiguous
    from fpc_amd.dist import shard_range
    got = [shard_range(250, 8, r_) for r_ in range(8)]
    assert got[0] == (0, 32) and got[1] == (32, 64) and got[2] == (64, 95) and got[-1] == (219, 250)
    assert all(a[1] == b[0] for a, b in zip(got, got[1:]))


def test_bench_parent_reports_a_failing_rank():
    """Without a GPU every rank fails when it creates its engine: the parent must come back promptly with a non-zero
    status (not wait in a collective) and print no result line."""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    t0 = __import__("time").time()
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], FPC_DIST_BACKEND="gloo")
    assert r.returncode != 0
    assert "a rank exited with status" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert __import__("time").time() - t0 < 120


def test_the_blob_collective_targets_the_librarys_buffer_and_falls_back_without_a_host_hop():
    """dist.nccl_receive_target with a stand-in engine (no GPU here; the real one is tests/test_gpu_round5.py): the tensor
    handed to the blob collective IS the engine's packed view when one can be made (then `finish` has the library verify
    the tag), and a torch-owned buffer handed over with import_packed_device when not (or FPC_DIST_ZERO_COPY=0) -- in
    neither case does a receiving rank go through export_packed / import_packed (host memory)."""
    from fpc_amd import dist as fdist

    class Fake:
        def __init__(self, n, view_ok=True):
            self.blob = torch.zeros(n, dtype=torch.uint8)
            self.view_ok = view_ok
            self.calls = []

        def packed_view(self):
            if not self.view_ok:
                raise RuntimeError("no view")
            return self.blob

        def mark_weights_loaded(self):
            self.calls.append("mark")

        def import_packed_device(self, buf):
            self.calls.append("import_device")
            self.blob.copy_(buf)

        def export_packed(self):
            self.calls.append("export")
            return self.blob.numpy()

        def import_packed(self, b):
            self.calls.append("import_host")

    n = 4096
    payload = torch.arange(n, dtype=torch.int64).to(torch.uint8)
    e = Fake(n)
    t, fin = fdist.nccl_receive_target(e, False, n, torch.device("cpu"))
    assert t.data_ptr() == e.blob.data_ptr()
    t.copy_(payload)
    fin()
    assert e.calls == ["mark"] and torch.equal(e.blob, payload)
    for env, view_ok in (("0", True), ("1", False)):
        os.environ["FPC_DIST_ZERO_COPY"] = env
        try:
            e = Fake(n, view_ok)
            t, fin = fdist.nccl_receive_target(e, False, n, torch.device("cpu"))
            assert t.data_ptr() != e.blob.data_ptr() and t.numel() == n
            t.copy_(payload)
            fin()
            assert e.calls == ["import_device"] and torch.equal(e.blob, payload)
            # the source rank's fallback: its blob into the torch buffer (the one place export_packed is allowed)
            s = Fake(n, view_ok)
            s.blob.copy_(payload)
            t, fin = fdist.nccl_receive_target(s, True, n, torch.device("cpu"))
            fin()
            assert torch.equal(t, payload) and "import_host" not in s.calls and "import_device" not in s.calls
        finally:
            del os.environ["FPC_DIST_ZERO_COPY"]
