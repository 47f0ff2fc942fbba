"""`Engine`: one fpc_ctx (one GPU, one frame geometry) behind a small Python class.

PyTorch is used for what it is good at here -- owning device memory and streams;
all arithmetic happens inside libfpc.so.  Tensors cross the boundary as raw
device pointers (`tensor.data_ptr()`).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, arch


def _as_tensor_table(state_dict):
    """{name: array-like} -> (FpcTensor[n], keep-alive list).  Only float entries go in
    (`num_batches_tracked` is an int64 counter the path never reads)."""
    keep, items = [], []
    for name, v in state_dict.items():
        if isinstance(v, torch.Tensor):
            if not v.dtype.is_floating_point:
                continue
            v = v.detach().cpu().contiguous().float().numpy()
        else:
            v = np.asarray(v)
            if v.dtype.kind != "f":
                continue
            v = np.ascontiguousarray(v, dtype=np.float32)
        if v.ndim > 4:
            raise ValueError("tensor %s has %d dims" % (name, v.ndim))
        keep.append(v)
        items.append((name.encode(), v))
    table = (_lib.FpcTensor * len(items))()
    for i, (name, v) in enumerate(items):
        table[i].name = name
        table[i].data = v.ctypes.data
        table[i].ndim = v.ndim
        for d in range(v.ndim):
            table[i].shape[d] = v.shape[d]
    keep.append(items)
    return table, keep


class Engine:
    def __init__(self, height, width, max_batch=1, device=0, nms_dist=4, conf_thresh=0.015,
                 border_remove=4, descriptor_enabled=True, max_keypoints=0, in_channels=3, dtype="f32", arch="resnet",
                 num_streams=0, plan_flags=(), nms_round_launches=0, min_sub_batch=0):
        self._l = _lib.load()          # raises if libfpc.so is not built: no fallback
        if not torch.cuda.is_available():
            raise RuntimeError("fpc_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU path in the product")
        cfg = _lib.FpcConfig()
        _lib.check(self._l.fpc_default_config(ctypes.byref(cfg)), "fpc_default_config")
        cfg.device, cfg.height, cfg.width, cfg.max_batch = device, height, width, max_batch
        cfg.nms_dist, cfg.conf_thresh, cfg.border_remove = nms_dist, conf_thresh, border_remove
        cfg.descriptor_enabled, cfg.max_keypoints = int(bool(descriptor_enabled)), max_keypoints
        cfg.in_channels = in_channels
        codes = {"f32": 0, "bf16": 1, "f32_split": 2, "f32_split_f16": 3}   # FPC_F32 ... FPC_F32_SPLIT_F16 (include/fpc.h)
        if dtype not in codes:
            raise ValueError("dtype must be one of %s, got %r" % (sorted(codes), dtype))
        cfg.dtype = codes[dtype]
        self.dtype = dtype
        if arch not in ("resnet", "vgg"):
            raise ValueError("arch must be 'resnet' (python/src/superpoint.py) or 'vgg' (cpp/src/model.cc), got %r" % (arch,))
        cfg.arch = 1 if arch == "vgg" else 0     # FPC_ARCH_RESNET / FPC_ARCH_VGG
        self.arch = arch
        self.in_channels = 1 if in_channels == 1 else 3
        # launch-plan knobs (fpc_config.num_streams / plan_flags / ...; zeros = the default plan)
        cfg.num_streams, cfg.nms_round_launches, cfg.min_sub_batch = num_streams, nms_round_launches, min_sub_batch
        flags = 0
        for f in ([plan_flags] if isinstance(plan_flags, (str, int)) else plan_flags):
            flags |= f if isinstance(f, int) else _lib.PLAN_FLAGS[f]
        cfg.plan_flags = flags
        self.cfg = cfg
        self.h, self.w, self.max_batch, self.device = height, width, max_batch, device
        self.descriptor_enabled = bool(descriptor_enabled)
        self._ctx = ctypes.c_void_p()
        _lib.check(self._l.fpc_create(ctypes.byref(self._ctx), ctypes.byref(cfg)), "fpc_create")
        self.torch_device = torch.device("cuda", device)
        res = _lib.FpcDeviceResults()
        _lib.check(self._l.fpc_results(self._ctx, ctypes.byref(res)), "fpc_results")
        self.capacity, self.desc_dim = res.capacity, res.desc_dim
        self._res = res

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            bad = 0
            if os.environ.get("FPC_GUARD_ZONES") == "1":
                # a whole run under the canary zones (FPC_GUARD_ZONES=1 python -m pytest tests -m gpu): every context is
                # checked when it is closed, and a kernel that stored outside its tensors fails the test that closed it
                b = ctypes.c_longlong(0)
                if self._l.fpc_check_guards(self._ctx, ctypes.byref(b)) == 0:
                    bad = int(b.value)
            self._l.fpc_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()
            if bad:
                raise RuntimeError("fpc_check_guards: %d canary words were overwritten -- a kernel stored outside its tensors" % bad)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Strict load of ckpt['model_state_dict'] (reference: saveutils.py:6-18)."""
        table, keep = _as_tensor_table(state_dict)
        _lib.check(self._l.fpc_load_weights(self._ctx, table, len(table)), "fpc_load_weights")
        del keep

    def packed_size(self):
        return self._l.fpc_packed_size(self._ctx)

    def packed_view(self):
        """The packed weight blob as a uint8 CUDA tensor aliasing the library's buffer
        (the in-place target of the RCCL broadcast, see dist.py)."""
        n = self.packed_size()
        ptr = self._l.fpc_packed_device_ptr(self._ctx)
        holder = _DevArray(ptr, n)
        return torch.as_tensor(holder, device=self.torch_device)

    def export_packed(self):
        buf = np.empty(self.packed_size(), np.uint8)
        _lib.check(self._l.fpc_export_packed(self._ctx, buf.ctypes.data, buf.nbytes), "fpc_export_packed")
        return buf

    def import_packed(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        _lib.check(self._l.fpc_import_packed(self._ctx, buf.ctypes.data, buf.nbytes), "fpc_import_packed")

    def import_packed_device(self, buf):
        """fpc_import_packed_device: a packed blob that is already in device memory (a uint8 CUDA tensor -- the receive
        buffer of a broadcast, another engine's packed_view()): tag checked, copied device to device."""
        if buf.dtype != torch.uint8 or not buf.is_cuda or not buf.is_contiguous():
            raise ValueError("import_packed_device takes a contiguous uint8 CUDA tensor")
        torch.cuda.synchronize(buf.device)
        _lib.check(self._l.fpc_import_packed_device(self._ctx, buf.data_ptr(), buf.numel()), "fpc_import_packed_device")

    def mark_weights_loaded(self):
        _lib.check(self._l.fpc_mark_weights_loaded(self._ctx), "fpc_mark_weights_loaded")

    def plan_hash(self):
        """Equal on two engines iff they can exchange packed weights (same build, dtype, arch, launch plan)."""
        return int(self._l.fpc_plan_hash(self._ctx))

    def broadcast_weights(self, nccl_comm, root=0):
        """fpc_broadcast_weights: RCCL broadcast of the packed blob on a raw ncclComm_t (an int / c_void_p)."""
        _lib.check(self._l.fpc_broadcast_weights(self._ctx, ctypes.c_void_p(nccl_comm), root), "fpc_broadcast_weights")

    # -- execution -------------------------------------------------------------------
    def use_torch_stream(self):
        s = torch.cuda.current_stream(self.torch_device).cuda_stream
        _lib.check(self._l.fpc_set_stream(self._ctx, ctypes.c_void_p(s)), "fpc_set_stream")

    def torch_stream(self):
        """The ctx's main stream as a torch stream (events recorded / awaited on it order torch work against fpc_detect)."""
        return torch.cuda.ExternalStream(int(self._l.fpc_get_stream(self._ctx)), device=self.torch_device)

    def upload_stream(self):
        """fpc_upload_stream: a stream for the caller's uploads that shares no hardware queue with the ctx's compute streams."""
        s = self._l.fpc_upload_stream(self._ctx)
        if not s:
            raise RuntimeError("fpc_upload_stream failed")
        return torch.cuda.ExternalStream(int(s), device=self.torch_device)

    def stream_report(self):
        """fpc_stream_report: which hardware queue each of the ctx's streams sits on, and what finding that out cost."""
        r = _lib.FpcStreamReport()
        _lib.check(self._l.fpc_stream_report(self._ctx, ctypes.byref(r)), "fpc_stream_report")
        names = {0: "main", 200: "upload"}
        streams = {}
        for i in range(r.n_streams):
            sl = r.slot[i]
            name = names.get(sl) or ("sub%d" % sl if sl < 100 else "side%d" % (sl - 100))
            streams[name] = r.queue[i]
        return {"streams": streams, "probing": bool(r.probing), "hw_queues_found": r.hw_queues_found,
                "create_probe_rounds": r.create_probe_rounds, "create_placement_ms": round(r.create_placement_ms, 3),
                "process_probe_rounds": r.process_probe_rounds, "process_probe_launches": r.process_probe_launches,
                "process_probe_ms": round(r.process_probe_ms, 3),
                "process_inconclusive_rounds": r.process_inconclusive_rounds,
                "process_registered_streams": r.process_registered_streams}

    def _frames(self, frames):
        if not isinstance(frames, torch.Tensor):
            frames = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32))
        frames = frames.to(self.torch_device, torch.float32).contiguous()
        if frames.dim() != 4 or frames.shape[1] != self.in_channels or frames.shape[2] != self.h or frames.shape[3] != self.w:
            raise ValueError("frames must be [n,%d,%d,%d], got %s" % (self.in_channels, self.h, self.w, tuple(frames.shape)))
        if frames.shape[0] > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (frames.shape[0], self.max_batch))
        return frames

    def forward(self, frames):
        """SuperPoint.forward (superpoint.py:91-115): -> prob_map [n,H,W], desc [n,128,H/8,W/8],
        logits [n,65,H/8,W/8] as CUDA tensors."""
        frames = self._frames(frames)
        n = frames.shape[0]
        dev = self.torch_device
        prob = torch.empty((n, self.h, self.w), device=dev)
        desc = torch.empty((n, self.desc_dim, self.h // 8, self.w // 8), device=dev)
        logits = torch.empty((n, 65, self.h // 8, self.w // 8), device=dev)
        torch.cuda.synchronize(dev)
        _lib.check(self._l.fpc_forward(self._ctx, frames.data_ptr(), n, prob.data_ptr(), desc.data_ptr(),
                                       logits.data_ptr()), "fpc_forward")
        self.sync()
        return prob, desc, logits

    def detect_async(self, frames_dev, n):
        """Enqueue the whole path for n device-resident frames (no sync, no copies)."""
        _lib.check(self._l.fpc_detect(self._ctx, frames_dev.data_ptr(), n), "fpc_detect")

    def detect(self, frames):
        frames = self._frames(frames)
        torch.cuda.synchronize(self.torch_device)
        self.detect_async(frames, frames.shape[0])
        return self.fetch(frames.shape[0])

    U8_LAYOUTS = {"gray": 0, "rgb_hwc": 1, "bgr_hwc": 2, "bgr_hwc_gray": 3}   # FPC_U8_* (include/fpc.h)

    def detect_u8_async(self, frames_u8_dev, n, layout):
        """8-bit device frames (camera.py:31: float32(u8) / 255.0, converted on the device) -> the whole path."""
        _lib.check(self._l.fpc_detect_u8(self._ctx, frames_u8_dev.data_ptr(), n, self.U8_LAYOUTS[layout]), "fpc_detect_u8")

    def detect_u8(self, frames_u8, layout):
        """frames_u8: uint8 [n,H,W] ("gray") or [n,H,W,3] ("rgb_hwc", "bgr_hwc", "bgr_hwc_gray")."""
        if not isinstance(frames_u8, torch.Tensor):
            frames_u8 = torch.from_numpy(np.ascontiguousarray(frames_u8, dtype=np.uint8))
        f = frames_u8.to(self.torch_device, torch.uint8).contiguous()
        want = (self.h, self.w) if layout == "gray" else (self.h, self.w, 3)
        if tuple(f.shape[1:]) != want or f.shape[0] > self.max_batch:
            raise ValueError("u8 frames must be [n<=%d,%s], got %s" % (self.max_batch, want, tuple(f.shape)))
        torch.cuda.synchronize(self.torch_device)
        self.detect_u8_async(f, f.shape[0], layout)
        return self.fetch(f.shape[0])

    def detect_u8_resized(self, frames_u8, layout="bgr_hwc"):
        """make_query_image (python/src/inference.py:72-85) + camera.py:31 on the device: uint8 [n,h,w,3] camera frames
        of any size -> resized to cover this engine's H x W, centre-cropped, converted, detected."""
        if not isinstance(frames_u8, torch.Tensor):
            frames_u8 = torch.from_numpy(np.ascontiguousarray(frames_u8, dtype=np.uint8))
        f = frames_u8.to(self.torch_device, torch.uint8).contiguous()
        if f.dim() != 4 or f.shape[3] != 3 or f.shape[0] > self.max_batch or layout not in ("rgb_hwc", "bgr_hwc"):
            raise ValueError("frames must be uint8 [n<=%d,h,w,3] in rgb_hwc / bgr_hwc layout" % self.max_batch)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_detect_u8_resized(self._ctx, f.data_ptr(), f.shape[0], f.shape[1], f.shape[2],
                                                 self.U8_LAYOUTS[layout]), "fpc_detect_u8_resized")
        return self.fetch(f.shape[0])

    def u8_staging(self, n):
        """The float frames [n,C,H,W] the last detect_u8 call fed to the network (a copy)."""
        self.sync()
        ptr = self._l.fpc_u8_staging(self._ctx)
        nbytes = n * self.in_channels * self.h * self.w * 4
        raw = torch.as_tensor(_DevArray(ptr, nbytes), device=self.torch_device)   # uint8 view of the library's buffer
        return raw.view(torch.float32).reshape(n, self.in_channels, self.h, self.w).clone()

    def homography_adaptation(self, frames, homographies, inverses=None, erosion_radius=8, aggregation="sum"):
        """homography_adaptation (python/src/homographies.py:250-324): frames [n,C,H,W], homographies [num,8] (flat,
        the convention of sample_homography) -> aggregated probability maps [n,H,W] (CUDA tensor)."""
        frames = self._frames(frames)
        hs = np.ascontiguousarray(homographies, np.float32).reshape(-1, 8)
        inv = None if inverses is None else np.ascontiguousarray(inverses, np.float32).reshape(-1, 8)
        if inv is not None and inv.shape != hs.shape:
            raise ValueError("inverses must match homographies")
        if aggregation not in ("sum", "max"):
            raise ValueError("Unknown aggregation method: %s" % aggregation)        # homographies.py:322
        out = torch.empty((frames.shape[0], self.h, self.w), device=self.torch_device)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_homography_adaptation(
            self._ctx, frames.data_ptr(), frames.shape[0], hs.ctypes.data, None if inv is None else inv.ctypes.data,
            hs.shape[0], int(erosion_radius), 1 if aggregation == "max" else 0, out.data_ptr()), "fpc_homography_adaptation")
        self.sync()
        return out

    def get_points(self, prob_map, desc_map=None):
        """Post-processing only, on caller-provided dense maps (netutils.py:78-121)."""
        prob_map = prob_map.to(self.torch_device, torch.float32).contiguous()
        n = prob_map.shape[0]
        dptr = None
        if desc_map is not None:
            desc_map = desc_map.to(self.torch_device, torch.float32).contiguous()
            dptr = desc_map.data_ptr()
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_get_points(self._ctx, prob_map.data_ptr(), dptr, n), "fpc_get_points")
        return self.fetch(n, with_desc=desc_map is not None)

    def sample_descriptors(self, desc_map, xy):
        """get_descriptors on its own (netutils.py:103-121): desc_map [D,H/8,W/8] (or [1,D,H/8,W/8]), xy float64 [K,2]
        (x, y) in pixels -> unit-norm descriptors float32 [K,D] (numpy)."""
        if not isinstance(desc_map, torch.Tensor):
            desc_map = torch.from_numpy(np.ascontiguousarray(desc_map, dtype=np.float32))
        dm = desc_map.to(self.torch_device, torch.float32).contiguous()
        if dm.dim() == 4 and dm.shape[0] == 1:
            dm = dm[0]
        if tuple(dm.shape) != (self.desc_dim, self.h // 8, self.w // 8):
            raise ValueError("descriptor map must be [%d,%d,%d], got %s" % (self.desc_dim, self.h // 8, self.w // 8, tuple(dm.shape)))
        pts = torch.from_numpy(np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)).to(self.torch_device)
        k = pts.shape[0]
        out = torch.empty((k, self.desc_dim), dtype=torch.float32, device=self.torch_device)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_sample_descriptors(self._ctx, dm.data_ptr(), pts.data_ptr(), k, out.data_ptr()),
                   "fpc_sample_descriptors")
        self.sync()
        return out.cpu().numpy()

    ACTIVATIONS = ("pool", "layer1.0", "layer1.1", "layer2.0", "layer2.1", "det.0", "det.1", "desc_in.0", "desc_in.1",
                   "up", "desc_out.0", "desc_out.1")

    def activation(self, name, frame0=0, n=1):
        """Intermediate tensor of the LAST forward / detect call as a forward hook on the reference module would return
        it: float32 CUDA tensor [n,C,h,w] (fpc_read_activation).  "det.1" (the logits) raises after a `detect` in
        dtype="bf16" with the fused softmax epilogue -- that call never writes logits; `forward` always does."""
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(self._l.fpc_read_activation(self._ctx, name.encode(), frame0, n, None, ctypes.byref(c), ctypes.byref(h),
                                               ctypes.byref(w)), "fpc_read_activation")
        out = torch.empty((n, c.value, h.value, w.value), dtype=torch.float32, device=self.torch_device)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_read_activation(self._ctx, name.encode(), frame0, n, out.data_ptr(), None, None, None),
                   "fpc_read_activation")
        self.sync()
        return out

    def sync(self):
        _lib.check(self._l.fpc_sync(self._ctx), "fpc_sync")

    def counts(self, n, allow_nonfinite=False):
        """fpc_get_counts.  A frame with a NaN / Inf pixel raises FpcError (code -9, FPC_E_NONFINITE) unless
        `allow_nonfinite`: the counts are delivered either way, and output_range() names the frame."""
        cnt = np.zeros(n, np.int32)
        ncand = np.zeros(n, np.int32)
        rc = self._l.fpc_get_counts(self._ctx, n, cnt.ctypes.data, ncand.ctypes.data)
        if not (allow_nonfinite and rc == -9):
            _lib.check(rc, "fpc_get_counts")
        return cnt, ncand

    def output_range(self, n):
        """fpc_output_range: per frame of the last call (max logit, max |descriptor map| (forward only), non-finite pixel?)."""
        ml, md, bad = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        _lib.check(self._l.fpc_output_range(self._ctx, n, ml.ctypes.data, md.ctypes.data, bad.ctypes.data), "fpc_output_range")
        return ml, md, bad.astype(bool)

    def fetch(self, n, with_desc=None, allow_nonfinite=False):
        """-> list of (xy int32[K,2], conf float32[K], desc float32[K,128] | None, n_candidates)."""
        if with_desc is None:
            with_desc = self.descriptor_enabled
        cnt, ncand = self.counts(n, allow_nonfinite)
        out = []
        for f in range(n):
            k = int(cnt[f])
            xy = np.empty((k, 2), np.int32)
            conf = np.empty(k, np.float32)
            desc = np.empty((k, self.desc_dim), np.float32) if with_desc else None
            got = self._l.fpc_get_keypoints(self._ctx, f, k, xy.ctypes.data, conf.ctypes.data,
                                            desc.ctypes.data if with_desc else None)
            _lib.check(got, "fpc_get_keypoints")
            assert got == k
            out.append((xy, conf, desc, int(ncand[f])))
        return out

    # -- descriptor matching (next row of the path) -------------------------------------
    def _desc(self, d):
        if not isinstance(d, torch.Tensor):
            d = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32))
        d = d.to(self.torch_device, torch.float32).contiguous()
        if d.dim() != 2 or d.shape[1] != self.desc_dim:
            raise ValueError("descriptors must be [n,%d]" % self.desc_dim)
        return d

    def match(self, query, train, cross_check=True, max_dist=0.0):
        """cv2.BFMatcher(NORM_L2, crossCheck).match(query, train): -> (match int32[nq] (train index or
        -1), dist float32[nq])."""
        q, t = self._desc(query), self._desc(train)
        m = torch.empty(q.shape[0], dtype=torch.int32, device=self.torch_device)
        d = torch.empty(q.shape[0], dtype=torch.float32, device=self.torch_device)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_match(self._ctx, q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0],
                                     int(bool(cross_check)), float(max_dist), m.data_ptr(), d.data_ptr()), "fpc_match")
        self.sync()
        return m.cpu().numpy(), d.cpu().numpy()

    def first_within(self, key, cur, tolerance=0.8):
        """SearchKeyFrameCorrespondence (cpp/src/main.cc:18-29): first index in `cur` closer than tolerance."""
        k, c = self._desc(key), self._desc(cur)
        out = torch.empty(k.shape[0], dtype=torch.int32, device=self.torch_device)
        torch.cuda.synchronize(self.torch_device)
        _lib.check(self._l.fpc_first_within(self._ctx, k.data_ptr(), k.shape[0], c.data_ptr(), c.shape[0],
                                            float(tolerance), out.data_ptr()), "fpc_first_within")
        self.sync()
        return out.cpu().numpy()

    # -- timing ----------------------------------------------------------------------
    def check_guards(self):
        """Contexts created with plan_flags=["guard_zones"] (a test facility): waits for the device and returns the number
        of 32-bit words of the workspace's canary zones that a kernel has overwritten (0 = every store stayed inside
        its tensor)."""
        bad = ctypes.c_longlong(0)
        _lib.check(self._l.fpc_check_guards(self._ctx, ctypes.byref(bad)), "fpc_check_guards")
        return int(bad.value)

    def set_timing(self, on):
        """on = True / False, or an int n > 1: only every n-th detect call (the first included) carries the events."""
        _lib.check(self._l.fpc_set_timing(self._ctx, int(on) if on is not True else 1), "fpc_set_timing")

    def timings(self):
        """[(layer name, kernel symbol, ms, algorithmic FLOPs, MFMA-issued FLOPs, algorithmic HBM bytes)] of the
        launches recorded since set_timing(True) (FLOPs and bytes are per LAUNCH: per frame x the launch's frames)."""
        cap = max(128, _lib.check(self._l.fpc_get_timings(self._ctx, 0, None, None, None, None, None, None), "fpc_get_timings"))
        names = (ctypes.c_char_p * cap)()
        kernels = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        fl = (ctypes.c_double * cap)()
        mf = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        n = _lib.check(self._l.fpc_get_timings(self._ctx, cap, names, kernels, ms, fl, mf, by), "fpc_get_timings")
        return [(names[i].decode(), kernels[i].decode(), float(ms[i]), float(fl[i]), float(mf[i]), float(by[i]))
                for i in range(min(n, cap))]


    def kernel_names(self, frames):
        """Kernel symbols one fpc_forward over `frames` launches (through the timing facility: events on, one call, events off)."""
        self.set_timing(True)
        self.forward(frames)
        self.sync()
        out = [k for _, k, _, _, _, _ in self.timings()]
        self.set_timing(False)
        return out


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can alias library-owned memory."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def conv_flops_per_frame(h, w, descriptor=True):
    return 2.0 * arch.conv_macs(h, w, descriptor)
