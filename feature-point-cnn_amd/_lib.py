"""ctypes binding of the C-ABI in include/fpc.h (libfpc.so, built in-tree by
`__graft_entry__.build()` / `make -C feature-point-cnn_amd/csrc`).

There is no fallback: if the shared library is missing or cannot be loaded this
module raises, and so does every entry point of the package.
"""
import ctypes
import os
import subprocess

# PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
# Import it BEFORE libfpc.so is mapped so that the loader resolves libfpc's
# libamdhip64.so.7 dependency to that already-loaded copy: two HIP runtimes in one
# process do not share a device.  (A C/C++ host links the system ROCm instead.)
import torch  # noqa: F401

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPC_LIB_PATH") or os.path.join(_DIR, "lib", "libfpc.so")   # FPC_LIB_PATH: A/B builds
CSRC = os.path.join(_DIR, "csrc")

# every symbol include/fpc.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "fpc_abi_version", "fpc_build_flags", "fpc_strerror", "fpc_last_hip_error", "fpc_default_config", "fpc_create",
    "fpc_destroy", "fpc_load_weights", "fpc_packed_size", "fpc_packed_device_ptr",
    "fpc_export_packed", "fpc_import_packed", "fpc_import_packed_device", "fpc_mark_weights_loaded", "fpc_set_stream", "fpc_upload_stream",
    "fpc_get_stream", "fpc_sync", "fpc_forward", "fpc_detect", "fpc_get_points", "fpc_results",
    "fpc_get_counts", "fpc_get_keypoints", "fpc_set_timing", "fpc_get_timings", "fpc_match", "fpc_first_within",
    "fpc_detect_u8", "fpc_u8_staging", "fpc_homography_adaptation", "fpc_detect_u8_resized",
    "fpc_sample_descriptors", "fpc_plan_hash", "fpc_broadcast_weights", "fpc_read_activation",
    "fpc_pack_layout_revision", "fpc_check_guards", "fpc_stream_report", "fpc_output_range",
]

ABI_VERSION = 4
PACK_LAYOUT_REVISION = 4      # include/fpc.h FPC_PACK_LAYOUT_REVISION

# fpc_config.plan_flags (include/fpc.h, FPC_PLAN_*)
PLAN_FLAGS = {
    "no_fused_blocks": 1 << 0, "no_winograd": 1 << 1, "no_winograd_detector": 1 << 2, "no_winograd_layer_in1": 1 << 3,
    "no_xcd_order": 1 << 4, "no_fused_stem_pool": 1 << 5, "split_heads": 1 << 6, "nms_in_line": 1 << 7,
    "no_persistent_grid": 1 << 8, "layer1_tile_8x16": 1 << 9, "winograd_gen1": 1 << 10, "no_latency_tiles": 1 << 11,
    "nms_one_workgroup": 1 << 12, "no_fused_softmax": 1 << 13, "winograd_gen2": 1 << 14, "guard_zones": 1 << 15, "detector_gen1": 1 << 16, "heads_in_line": 1 << 17, "w36_one_wave": 1 << 18, "convt_phases": 1 << 19, "stem_round3": 1 << 20, "conv_round1": 1 << 21,
}


class FpcConfig(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("height", ctypes.c_int), ("width", ctypes.c_int),
                ("max_batch", ctypes.c_int), ("cell", ctypes.c_int), ("nms_dist", ctypes.c_int),
                ("conf_thresh", ctypes.c_float), ("border_remove", ctypes.c_int),
                ("descriptor_enabled", ctypes.c_int), ("max_keypoints", ctypes.c_int),
                ("in_channels", ctypes.c_int), ("dtype", ctypes.c_int), ("arch", ctypes.c_int),
                ("num_streams", ctypes.c_int), ("plan_flags", ctypes.c_uint), ("nms_round_launches", ctypes.c_int),
                ("min_sub_batch", ctypes.c_int)]


class FpcTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int),
                ("shape", ctypes.c_int64 * 4)]


class FpcDeviceResults(ctypes.Structure):
    _fields_ = [("count", ctypes.c_void_p), ("n_candidates", ctypes.c_void_p), ("xy", ctypes.c_void_p),
                ("conf", ctypes.c_void_p), ("desc", ctypes.c_void_p), ("capacity", ctypes.c_int),
                ("desc_dim", ctypes.c_int)]


class FpcStreamReport(ctypes.Structure):
    _fields_ = [("n_streams", ctypes.c_int), ("slot", ctypes.c_int * 16), ("queue", ctypes.c_int * 16),
                ("probing", ctypes.c_int), ("hw_queues_found", ctypes.c_int), ("process_probe_rounds", ctypes.c_int),
                ("process_probe_launches", ctypes.c_int), ("process_inconclusive_rounds", ctypes.c_int),
                ("process_probe_ms", ctypes.c_float), ("create_probe_rounds", ctypes.c_int),
                ("create_placement_ms", ctypes.c_float), ("process_registered_streams", ctypes.c_int)]


class FpcError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        l = load()
        msg = l.fpc_strerror(code).decode()
        detail = l.fpc_last_hip_error().decode()
        super().__init__("%s: %s (%d)%s" % (where, msg, code, " -- " + detail if detail else ""))


_lib = None


def build(force=False):
    """Compile libfpc.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libfpc.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C feature-point-cnn_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    l.fpc_abi_version.restype = ci
    l.fpc_build_flags.restype = ctypes.c_char_p
    l.fpc_strerror.restype = ctypes.c_char_p
    l.fpc_strerror.argtypes = [ci]
    l.fpc_last_hip_error.restype = ctypes.c_char_p
    l.fpc_default_config.argtypes = [ctypes.POINTER(FpcConfig)]
    l.fpc_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(FpcConfig)]
    l.fpc_destroy.argtypes = [vp]
    l.fpc_destroy.restype = None
    l.fpc_load_weights.argtypes = [vp, ctypes.POINTER(FpcTensor), ci]
    l.fpc_packed_size.argtypes = [vp]
    l.fpc_packed_size.restype = ctypes.c_size_t
    l.fpc_packed_device_ptr.argtypes = [vp]
    l.fpc_packed_device_ptr.restype = vp
    l.fpc_export_packed.argtypes = [vp, vp, ctypes.c_size_t]
    l.fpc_import_packed.argtypes = [vp, vp, ctypes.c_size_t]
    l.fpc_import_packed_device.argtypes = [vp, vp, ctypes.c_size_t]
    l.fpc_mark_weights_loaded.argtypes = [vp]
    l.fpc_set_stream.argtypes = [vp, vp]
    l.fpc_upload_stream.argtypes = [vp]
    l.fpc_upload_stream.restype = vp
    l.fpc_get_stream.argtypes = [vp]
    l.fpc_get_stream.restype = vp
    l.fpc_sync.argtypes = [vp]
    l.fpc_forward.argtypes = [vp, vp, ci, vp, vp, vp]
    l.fpc_detect.argtypes = [vp, vp, ci]
    l.fpc_get_points.argtypes = [vp, vp, vp, ci]
    l.fpc_detect_u8.argtypes = [vp, vp, ci, ci]
    l.fpc_u8_staging.argtypes = [vp]
    l.fpc_detect_u8_resized.argtypes = [vp, vp, ci, ci, ci, ci]
    l.fpc_homography_adaptation.argtypes = [vp, vp, ci, vp, vp, ci, ci, ci, vp]
    l.fpc_u8_staging.restype = vp
    l.fpc_results.argtypes = [vp, ctypes.POINTER(FpcDeviceResults)]
    l.fpc_get_counts.argtypes = [vp, ci, vp, vp]
    l.fpc_get_keypoints.argtypes = [vp, ci, ci, vp, vp, vp]
    l.fpc_output_range.argtypes = [vp, ci, vp, vp, vp]
    l.fpc_match.argtypes = [vp, vp, ci, vp, ci, ci, ctypes.c_float, vp, vp]
    l.fpc_first_within.argtypes = [vp, vp, ci, vp, ci, ctypes.c_float, vp]
    l.fpc_sample_descriptors.argtypes = [vp, vp, vp, ci, vp]
    l.fpc_read_activation.argtypes = [vp, ctypes.c_char_p, ci, ci, vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    l.fpc_plan_hash.argtypes = [vp]
    l.fpc_plan_hash.restype = ctypes.c_uint64
    l.fpc_broadcast_weights.argtypes = [vp, vp, ci]
    l.fpc_pack_layout_revision.restype = ci
    l.fpc_check_guards.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    l.fpc_stream_report.argtypes = [vp, ctypes.POINTER(FpcStreamReport)]
    l.fpc_set_timing.argtypes = [vp, ci]
    l.fpc_get_timings.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                  ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    for s in SYMBOLS:
        getattr(l, s)  # AttributeError if the library lacks a declared symbol
    if l.fpc_abi_version() != ABI_VERSION:
        raise ImportError("libfpc.so has ABI version %d, this binding is for %d: rebuild (make -C %s)"
                          % (l.fpc_abi_version(), ABI_VERSION, CSRC))
    if l.fpc_pack_layout_revision() != PACK_LAYOUT_REVISION:
        raise ImportError("libfpc.so packs fragment-layout revision %d, this binding expects %d: rebuild (make -C %s)"
                          % (l.fpc_pack_layout_revision(), PACK_LAYOUT_REVISION, CSRC))
    _lib = l
    return l


def check(code, where):
    if code < 0:
        raise FpcError(code, where)
    return code
