/*
 * fpc.h -- C-ABI of the MI355X-native SuperPoint inference path (libfpc.so).
 *
 * The reference (Kolkir/feature-point-cnn) has no plugin / FFI interface for this
 * path: its boundary is a pair of plain classes,
 *     C++    superpoint::SuperPoint            cpp/src/superpoint.h:12-36
 *     Python InferenceWrapper / SuperPoint     python/src/inferencewrapper.py:12-46,
 *                                              python/src/superpoint.py:64-115
 * Each entry point below names the reference code it stands in for.  Plain
 * pointers and sizes only; no torch / OpenCV types cross this boundary.  Every
 * function returns FPC_OK (0) or a negative FPC_E_* code -- nothing exits or
 * throws across the ABI (the reference exit()s: cpp/src/superpoint.cc:56-59,
 * python/src/saveutils.py:11-14).
 *
 * Threading: one fpc_ctx per GPU; calls on one ctx are serialised by the caller
 * (the reference object is not re-entrant either: cpp/src/superpoint.h:31-35);
 * distinct contexts are independent.
 *
 * There is no CPU fallback: without a HIP device fpc_create fails with
 * FPC_E_NO_DEVICE.
 */
#ifndef FPC_H
#define FPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPC_ABI_VERSION 4
/* Revision of the packed-weight FRAGMENT LAYOUTS (what the packing routines of this build write at the offsets the launch
 * plan names).  It is part of the blob's tag and of fpc_plan_hash: bump it whenever a packing routine changes the order
 * of values inside a layer's fragments -- offsets and sizes, which the plan hash covers anyway, do not change then, and
 * a blob of the older build would otherwise be accepted and multiplied with the wrong weights.
 *   1: rounds 1-2.   2: round 3 (fp32 stem fragments in the column / row tap-pair K order, stem_pair).
 *   3: round 4.  (ABI 4, round 5: fpc_stream_report added, fpc_set_stream's contract narrowed.)
 *   4: round 5 (fp32 stem fragments carry the folded-BN bias as a K step where stem_pool2_kernel runs; FPC_BF16's
 *      ConvTranspose fragments in convt_bf16_kernel's nine-taps-per-step order). */
#define FPC_PACK_LAYOUT_REVISION 4

enum {
  FPC_OK = 0,
  FPC_E_INVALID = -1,     /* bad argument / unsupported geometry                  */
  FPC_E_NO_DEVICE = -2,   /* no usable HIP device                                 */
  FPC_E_HIP = -3,         /* a HIP runtime call failed (fpc_last_hip_error)       */
  FPC_E_NO_WEIGHTS = -4,  /* forward/detect before any weights were loaded        */
  FPC_E_MISSING_KEY = -5, /* checkpoint entry missing or of the wrong shape       */
  FPC_E_CAPACITY = -6,    /* caller buffer too small (needed size is reported)    */
  FPC_E_NOT_CONVERGED = -7, /* NMS round limit hit (never seen)    */
  FPC_E_RANGE = -8,       /* FPC_F32_SPLIT_F16 only: a folded weight (at load) or an */
                          /* activation (reported by fpc_get_counts) left fp16's     */
                          /* range |x| <= 65504; use FPC_F32_SPLIT or FPC_F32        */
  FPC_E_NONFINITE = -9    /* fpc_get_counts: a frame of the call held a NaN or +-Inf */
                          /* pixel (see "Numerical contract" below)                  */
};

/* Numerical contract of FPC_F32 (the default), measured against a double-accumulating restatement of the reference on
 * checkpoints whose every activation was scaled x1 .. x100 (tests/test_gpu_round5.py):
 *  - the PRODUCTS of the path -- probability map, keypoint coordinates and confidences, unit-norm descriptors -- are
 *    within 1e-4 (measured <= 1.7e-5) / identical sets at every magnitude the reference itself survives: its softmax has
 *    no max-subtraction (python/src/superpoint.py:111-112), so logits above 88.7 overflow exp() there and here alike
 *    (NaN probabilities, no keypoint in such a cell);
 *  - the dense fp32 tensors fpc_forward returns (logits, descriptor map) carry fp32's RELATIVE error:
 *    |delta| <= 2.5e-6 * max|tensor| (measured 1.5e-6; the 3x3 layers run as Winograd F(4x4,3x3), whose F(2x2,3x3) and
 *    direct alternatives measure 1.0-1.7x smaller, i.e. the bound is fp32 accumulation over 12 layers, not the
 *    transform).  An ABSOLUTE 1e-4 therefore holds while max|tensor| <= 40 (13 ulp of fp32 at 64): fpc_output_range
 *    reports the magnitudes of the last call so that a caller who needs the absolute bar can check it;
 *  - frames must be finite.  FPC_F32's stem looks at every pixel it stages: a NaN or +-Inf pixel makes fpc_get_counts
 *    return FPC_E_NONFINITE for the call (after delivering the counts) and fpc_output_range name the frame.  The other
 *    frames of the batch are unaffected (frames are independent), and the call is memory-safe; the flagged frame's own
 *    results are UNDEFINED -- the reference propagates the NaN through the receptive field of the pixel (a region of
 *    NaN logits, no keypoints there), this library's ReLU (v_max_f32 returns the non-NaN operand) does not, and its
 *    Winograd tiles spread whatever survives further than a direct convolution would.  fpc_detect_u8* inputs are 8-bit
 *    and cannot be non-finite.  The other dtype modes do not check. */

enum { FPC_F32 = 0, FPC_BF16 = 1, FPC_F32_SPLIT = 2, FPC_F32_SPLIT_F16 = 3 };
enum { FPC_ARCH_RESNET = 0, FPC_ARCH_VGG = 1 };

/* Replaces SuperPointSettings (python/src/settings.py:2-8) / Settings
 * (cpp/src/settings.h:27-31) plus the geometry the reference takes from the frame. */
typedef struct fpc_config {
  int device;             /* HIP device ordinal                                    */
  int height, width;      /* frame size; multiples of 16 (8 if descriptor_enabled=0)*/
  int max_batch;          /* frames per fpc_detect / fpc_forward call              */
  int cell;               /* 8   settings.py:7   (only 8 is supported)             */
  int nms_dist;           /* 4   settings.py:4                                     */
  float conf_thresh;      /* 0.015 settings.py:5; finite and >= 0 (probabilities)  */
  int border_remove;      /* 4   settings.py:8                                     */
  int descriptor_enabled; /* 0 = MagicPoint, detector only (superpoint.py:103-109) */
  int max_keypoints;      /* per-frame output capacity; 0 = worst case for nms_dist; */
                          /* a smaller value keeps the most confident points        */
  int in_channels;        /* 0 or 3: frames [n,3,H,W] (superpoint.py:12); 1: gray frames   */
                          /* [n,1,H,W] -- what the reference feeds after replicating the   */
                          /* plane x3 (dataset_utils.py:19-20, cpp/src/camera.cc:17-18)    */
  int dtype;              /* FPC_F32 (0): fp32 activations and weights, the reference's    */
                          /* arithmetic; FPC_BF16 (1): bf16 activations / weights with     */
                          /* fp32 accumulation (BASELINE.json configs[4]); FPC_F32_SPLIT   */
                          /* (2): fp32 tensors, matrix products on the bf16 pipe with each */
                          /* operand split exactly into three bf16 terms (block_x3.h):     */
                          /* fp32-level accuracy, same 1e-4 parity bar as FPC_F32;         */
                          /* FPC_F32_SPLIT_F16 (3): the same with two fp16 terms per        */
                          /* operand and three MFMAs per product (faster; operands must lie */
                          /* in fp16's range, |x| <= 65504, or they saturate)               */
  int arch;               /* FPC_ARCH_RESNET (0): the Python network (superpoint.py), 128-D; */
                          /* FPC_ARCH_VGG (1): the C++ frontend's superpoint::SPModel        */
                          /* (cpp/src/model.cc, settings.h:19-25): gray input                */
                          /* (in_channels = 1), 256-D descriptors, fp32 or split mode        */
  /* Launch-plan knobs (zeros = the default plan).  The FPC_* environment variables of the
   * same names still override them, for A/B runs of an unmodified caller. */
  int num_streams;        /* sub-batches of one call on separate HIP streams; 0 = default (2; 3 in split modes)  */
  unsigned plan_flags;    /* FPC_PLAN_* bits                                                                   */
  int nms_round_launches; /* parallel NMS launches before the per-frame finish: 0 = default (2), -1 = none     */
  int min_sub_batch;      /* smallest sub-batch worth its own stream: 0 = default (8); calls below twice this   */
                          /* take the latency plan                                                             */
} fpc_config;

enum {
  FPC_PLAN_NO_FUSED_BLOCKS = 1 << 0,       /* conv1 / conv2 of a ResNetBlock as two launches            (FPC_FUSE=0)         */
  FPC_PLAN_NO_WINOGRAD = 1 << 1,           /* direct 3x3 convolutions everywhere                        (FPC_WINOGRAD=0)     */
  FPC_PLAN_NO_WINOGRAD_DETECTOR = 1 << 2,  /* ... in the detector's 65-channel blocks only              (FPC_WINOGRAD_DET=0) */
  FPC_PLAN_NO_WINOGRAD_LAYER_IN1 = 1 << 3, /* descriptor.layer_in.1 as the fused direct block           (FPC_WINOGRAD_IN1=0) */
  FPC_PLAN_NO_XCD_ORDER = 1 << 4,          /* plain tile order instead of the XCD-aware one             (FPC_XCD_ORDER=0)    */
  FPC_PLAN_NO_FUSED_STEM_POOL = 1 << 5,    /* stem convolution and max-pool as two launches             (FPC_FUSE_STEM=0)    */
  FPC_PLAN_SPLIT_HEADS = 1 << 6,           /* detector head + NMS on a side stream next to the descriptor head (FPC_SPLIT_HEADS=1): the
                                            * default, since round 4, of the Python network in FPC_F32 with two or more sub-batches */
  FPC_PLAN_NMS_IN_LINE = 1 << 7,           /* NMS on the sub-batch stream, not on a side stream         (FPC_NMS_ASIDE=0)    */
  FPC_PLAN_NO_PERSISTENT_GRID = 1 << 8,    /* one workgroup per tile in the Winograd kernels            (FPC_PERSIST_MIN=0)  */
  FPC_PLAN_LAYER1_TILE_8x16 = 1 << 9,      /* direct layer1 blocks on 8x16 instead of 16x16 tiles       (FPC_L1_T816=1)      */
  FPC_PLAN_WINOGRAD_GEN1 = 1 << 10,        /* round-1 Winograd kernel for the 64- / 128-channel layers  (FPC_WINOGRAD_GEN=1) */
  FPC_PLAN_NO_LATENCY_TILES = 1 << 11,     /* calls of a few frames keep the 8x16 tiles of the batch plan (FPC_LATENCY_TILES=0) */
  FPC_PLAN_NMS_ONE_WORKGROUP = 1 << 12,    /* survivors of a frame sorted by one workgroup, not in slices (FPC_NMS_CHUNKED=0)   */
  FPC_PLAN_NO_FUSED_SOFTMAX = 1 << 13,     /* FPC_BF16: exp-softmax as its own launch in fpc_detect too  (FPC_FUSE_SOFTMAX=0)   */
  FPC_PLAN_WINOGRAD_GEN2 = 1 << 14,        /* round-2 Winograd kernel, F(2x2,3x3), instead of F(4x4,3x3) (FPC_WINOGRAD_GEN=2)   */
  FPC_PLAN_HEADS_IN_LINE = 1 << 17,        /* ... and its opt-out: the two heads of a sub-batch back to back on its stream  (FPC_SPLIT_HEADS=0) */
  FPC_PLAN_DETECTOR_GEN1 = 1 << 16,        /* the detector's 65-channel blocks on round 1's kernel in batch calls too       (FPC_WINOGRAD_DET_GEN=1) */
  FPC_PLAN_W36_ONE_WAVE = 1 << 18,         /* the 64-channel F(4x4,3x3) layers on round 3's one-wave-per-SIMD kernel instead of round 5's two  (FPC_W36_PAIRED=0) */
  FPC_PLAN_STEM_ROUND3 = 1 << 20,          /* FPC_F32: round 3's stem_pool_kernel also where round 5's stem_pool2_kernel applies (conv map of whole 16x16 tiles) (FPC_STEM_LEAN=0) */
  FPC_PLAN_CONV_ROUND1 = 1 << 21,          /* FPC_F32: round 1's conv_mfma_kernel also where round 5's conv2_mfma_kernel applies (ConvTranspose, layer_in.1's 1x1) (FPC_CONV_LEAN=0) */
  FPC_PLAN_CONVT_PHASES = 1 << 19,         /* FPC_BF16: the ConvTranspose as four output-parity launches (rounds 2-4) instead of round 5's one  (FPC_CONVT_FUSED=0) */
  FPC_PLAN_GUARD_ZONES = 1 << 15           /* TEST FACILITY: 64 KiB of a canary pattern behind every buffer of the workspace and
                                              2 GiB behind the last one (the workspace grows by that much); fpc_check_guards
                                              counts the words a kernel has overwritten.  Not for production contexts.          */
};

/* One checkpoint entry: name and shape as in ckpt['model_state_dict']
 * (python/src/saveutils.py:57-62), data in host memory. */
typedef struct fpc_tensor {
  const char* name;
  const float* data;      /* float32, contiguous, PyTorch layout                   */
  int ndim;
  int64_t shape[4];
} fpc_tensor;

/* Device-resident results of the last fpc_detect call (zero-copy view). */
typedef struct fpc_device_results {
  const int32_t* count;       /* [n]            keypoints per frame (after border crop)    */
  const int32_t* n_candidates;/* [n]            pixels that passed conf_thresh             */
  const int32_t* xy;          /* [n][cap][2]    x, y                                       */
  const float* conf;          /* [n][cap]       descending (ties: row-major index ascending)*/
  const float* desc;          /* [n][cap][D]    unit L2 norm; NULL when descriptors are off */
  int capacity;               /* cap                                                       */
  int desc_dim;               /* D = 128 (256 for FPC_ARCH_VGG)                            */
} fpc_device_results;

typedef struct fpc_ctx fpc_ctx;

int fpc_abi_version(void);
/* "arch=gfx950;diag=0;ablations=none" for the product library: the code-object target, whether in-kernel stamps are
 * compiled in (the diagnostic build, never shipped), and that no experiment switch reached the build.  Static string. */
const char* fpc_build_flags(void);
const char* fpc_strerror(int code);
/* Text of the last HIP error seen by this thread (for FPC_E_HIP). */
const char* fpc_last_hip_error(void);

/* Fills the reference's defaults (settings.py:2-8), 480x640, max_batch 1. */
int fpc_default_config(fpc_config* cfg);

/* ~ SuperPoint::SuperPoint (cpp/src/superpoint.cc:9-66) / InferenceWrapper.__init__
 * (python/src/inferencewrapper.py:13-27) minus the file parsing: allocates the
 * device workspace for max_batch frames.  FPC_E_INVALID for a geometry the kernels cannot tile (height / width below
 * 16 or not a multiple of the network's stride, width above 3328, 2^30 pixels or more per frame) and for
 * max_batch * height * width >= 2^28 (the kernels address the tensors of a batch with 32-bit byte offsets, up to 16
 * bytes per frame pixel: 64 frames of 1280x960 are 2^26.2; larger jobs run as several calls or contexts). */
int fpc_create(fpc_ctx** out, const fpc_config* cfg);
void fpc_destroy(fpc_ctx* ctx);

/* ~ load_checkpoint_for_inference (python/src/saveutils.py:6-18), strict: every
 * learnable entry of table W must be present with the right shape
 * (`num_batches_tracked` entries are ignored).  Folds BatchNorm (eval mode,
 * eps 1e-5) into the convolutions and packs MFMA fragments on the device.
 * With descriptor_enabled == 0 the `descriptor.*` entries may be absent. */
int fpc_load_weights(fpc_ctx* ctx, const fpc_tensor* tensors, int n);

/* Packed, folded weights as one blob -- what rank 0 broadcasts over RCCL/xGMI
 * instead of every rank parsing the checkpoint.  Layout is private to one build. */
size_t fpc_packed_size(const fpc_ctx* ctx);
void* fpc_packed_device_ptr(fpc_ctx* ctx);              /* in-place collective target */
int fpc_export_packed(fpc_ctx* ctx, void* host_dst, size_t cap);
int fpc_import_packed(fpc_ctx* ctx, const void* host_src, size_t n);
/* The same from DEVICE memory (the receive buffer of a broadcast, another ctx's fpc_packed_device_ptr): the tag is read
 * back and checked, the blob is copied device to device.  Synchronous. */
int fpc_import_packed_device(fpc_ctx* ctx, const void* dev_src, size_t n);
/* Declares the blob at fpc_packed_device_ptr valid (after a broadcast into it).  The blob starts with a 64-byte tag
 * (magic, ABI version, dtype, arch, launch-plan hash, size); fpc_import_packed and this call refuse a blob whose tag
 * does not match the context (FPC_E_INVALID; fpc_last_hip_error says which field). */
int fpc_mark_weights_loaded(fpc_ctx* ctx);
/* Hash of everything the blob layout depends on -- equal on two contexts iff they can exchange packed weights. */
uint64_t fpc_plan_hash(const fpc_ctx* ctx);
/* FPC_PACK_LAYOUT_REVISION of the library as built (a binding compares it with the header it was written against). */
int fpc_pack_layout_revision(void);

/* TEST FACILITY (contexts created with FPC_PLAN_GUARD_ZONES only; FPC_E_INVALID otherwise): waits for the device, then
 * counts the 32-bit words of the workspace's canary zones -- behind every activation / result buffer, and 2 GiB behind
 * the last one -- that no longer hold the pattern written at fpc_create.  0 = no kernel stored outside its tensors. */
int fpc_check_guards(fpc_ctx* ctx, long long* bad_words);

/* Frame-batch sharding over the GPUs of a node (the heaviest batch caller to shard is
 * python/src/preprocess_coco.py:64-74): the ONE exchange of the path.  The root rank has loaded the checkpoint
 * (fpc_load_weights); every rank of the communicator calls this with its own ctx, and receives the packed, BN-folded
 * blob by ncclBroadcast (RCCL over xGMI) straight into its device buffer -- no other rank parses the file.
 * nccl_comm is an ncclComm_t (passed as void* so that this header needs no rccl.h); librccl.so is resolved at run
 * time (FPC_E_HIP if absent).  Two collectives on the ctx stream: the 64-byte tag, then the blob; a root without
 * weights makes every rank return FPC_E_NO_WEIGHTS after the first; a rank whose ctx (dtype, arch, plan, build)
 * differs from the root's still completes both collectives, then returns FPC_E_INVALID.  Synchronous. */
int fpc_broadcast_weights(fpc_ctx* ctx, void* nccl_comm, int root);

/* Work is enqueued on this hipStream_t (default: a stream the ctx owns).  A stream handed in must stay valid until
 * fpc_destroy or the next fpc_set_stream.  Handing over the stream the ctx already runs on returns at once (cheap enough
 * to do before every call).  A NEW caller stream is looked at once: if it is idle and not being captured into a graph,
 * one ~40 us one-thread kernel is launched on it to learn which hardware queue it sits on (so that the ctx's sub-batch
 * streams can keep clear of that queue); a busy or capturing stream, the null stream, or FPC_QUEUE_PROBE=0: nothing is
 * launched and the ctx's other streams stay as they are.  Never synchronises the caller's stream. */
int fpc_set_stream(fpc_ctx* ctx, void* hip_stream);
void* fpc_get_stream(fpc_ctx* ctx);
/* A hipStream_t for the CALLER's uploads (hipMemcpyAsync of the next batch while this one computes): non-blocking, owned
 * by the ctx (destroyed with it), made at the first call.  It is chosen -- by the probe fpc_create uses for the ctx's own
 * streams -- so that it does not share a hardware queue with the ctx's main / sub-batch streams nor, where a queue is
 * left, with those of the device's other live ctxs: an upload on a stream that shares a queue with compute waits behind
 * every launch in front of it.  Order it against fpc_detect with events, as any two streams.  NULL on failure.
 * (The reference uploads on the default stream: python/src/superpoint.py:98-99 `.cuda()`.) */
void* fpc_upload_stream(fpc_ctx* ctx);
/* How the ctx's streams were placed on the GPU's hardware queues, and what that cost (csrc/queue_map.h).  The HIP runtime
 * maps streams onto GPU_MAX_HW_QUEUES (4) queues and two streams on one queue run their kernels in a row, so fpc_create
 * picks streams that sit on queues of their own.  queue[i] of stream slot[i] (0 = main, 1.. = sub-batch streams, 100.. =
 * side streams, 200 = upload stream): index of the hardware queue, -1 = a queue outside the map, -3 = not probed
 * (FPC_QUEUE_PROBE=0, a caller stream that was busy, or probing switched off after inconclusive rounds: `probing` 0).
 * process_* count every probe round of the process on this device; create_* what THIS ctx's fpc_create spent. */
typedef struct fpc_stream_report_t {
  int n_streams;
  int slot[16], queue[16];
  int probing;
  int hw_queues_found;
  int process_probe_rounds, process_probe_launches, process_inconclusive_rounds;
  float process_probe_ms;
  int create_probe_rounds;
  float create_placement_ms;
  int process_registered_streams;   /* streams of ALL live ctxs of the process in the registry (0 once every ctx is gone) */
} fpc_stream_report_t;
int fpc_stream_report(fpc_ctx* ctx, fpc_stream_report_t* out);
int fpc_sync(fpc_ctx* ctx);

/* ~ SuperPoint.forward (python/src/superpoint.py:91-115): frames [n,3,H,W] ([n,1,H,W] with
 * in_channels = 1) float32
 * on the device -> prob_map [n,H,W], desc [n,128,H/8,W/8], logits [n,65,H/8,W/8]
 * (device, NCHW like the reference; any output may be NULL).  Asynchronous. */
int fpc_forward(fpc_ctx* ctx, const float* frames_dev, int n, float* prob_map_dev,
                float* desc_dev, float* logits_dev);

/* The intermediate tensors the last fpc_forward / fpc_detect left in the workspace -- what a forward hook on the
 * reference's modules returns -- for per-layer parity tests (SURVEY.md 8c fixture F1).  name: "pool" (encoder.max_pool),
 * "layer1.0", "layer1.1", "layer2.0", "layer2.1" (encoder blocks), "det.0", "det.1" (= logits), "desc_in.0",
 * "desc_in.1", "up" (descriptor.relu after the transposed convolution + bn), "desc_out.0", "desc_out.1" (= the
 * descriptor map).  Frames frame0 .. frame0+n-1 as float32 NCHW into out_dev [n,C,h,w] (bf16 tensors of the FPC_BF16
 * mode are widened exactly); channels / height / width receive the tensor's shape (out_dev may be NULL to query it).
 * The fused kernels never write a block's inner tensor h nor the un-pooled stem output to memory: those have no name
 * here.  "det.1" exists only when the last call produced logits: fpc_forward always does; fpc_detect does except in
 * FPC_BF16 with the fused softmax epilogue (the default there; FPC_PLAN_NO_FUSED_SOFTMAX turns it off), where the
 * name returns FPC_E_INVALID instead of stale memory.  FPC_ARCH_RESNET only.  Asynchronous on the ctx stream. */
int fpc_read_activation(fpc_ctx* ctx, const char* name, int frame0, int n, float* out_dev, int* channels, int* height,
                        int* width);

/* ~ InferenceWrapper.run (inferencewrapper.py:29-46) / SuperPoint::ProcessFrame
 * (cpp/src/superpoint.cc:68-96) for n independent frames: forward, exp-softmax,
 * depth-to-space, threshold, greedy NMS, sort, border crop (netutils.py:78-100,
 * nms.py:4-53), descriptor sampling + L2 normalisation (netutils.py:103-121).
 * Asynchronous; results stay on the device until fetched. */
int fpc_detect(fpc_ctx* ctx, const float* frames_dev, int n);

/* The step in front of the path, on the device: 8-bit camera frames -> the float frames
 * fpc_detect takes, then fpc_detect.  Uploading u8 instead of fp32 RGB cuts the host-to-device bytes 4x
 * (12x for gray).  Conversion is `float32(u8) / 255.0f` (python/src/camera.py:31, dataset_utils.py:23,
 * preprocess_coco.py:25), written as planar [n,C,H,W] into a staging buffer the ctx allocates on first use.
 *   FPC_U8_GRAY         [n,H,W]    -> [n,1,H,W]  (ctx with in_channels = 1)
 *   FPC_U8_RGB_HWC      [n,H,W,3]  -> [n,3,H,W]  (ctx with in_channels = 3; inferencewrapper.py:70-81)
 *   FPC_U8_BGR_HWC      [n,H,W,3]  -> [n,3,H,W] with the channel swap of cv2.COLOR_BGR2RGB (inference.py:79)
 *   FPC_U8_BGR_HWC_GRAY [n,H,W,3]  -> [n,1,H,W]  cv::COLOR_BGR2GRAY on 8-bit data then convertTo(CV_32FC1,
 *                       1/255) (cpp/src/camera.cc:17-18): OpenCV's documented 14-bit fixed-point weights
 *                       (B 1868, G 9617, R 4899, +8192 >> 14) and a float multiply by (float)(1.0/255.0).
 *                       OpenCV is not available to this build: parity for THIS layout is unpinned.
 * Resizing (cv2.resize, inference.py:72-85) is not done here. */
enum { FPC_U8_GRAY = 0, FPC_U8_RGB_HWC = 1, FPC_U8_BGR_HWC = 2, FPC_U8_BGR_HWC_GRAY = 3 };
int fpc_detect_u8(fpc_ctx* ctx, const uint8_t* frames_dev, int n, int layout);
/* Camera frames of another size: make_query_image (python/src/inference.py:72-85) on the device, fused with the
 * conversion above -- ratio-preserving bilinear resize so that the frame covers the ctx's H x W (new size
 * int(src * max(H/src_h, W/src_w)) per axis), centre crop, /255, optional BGR -> RGB, HWC -> CHW -- then fpc_detect.
 * frames_dev [n,src_h,src_w,3] u8; layout FPC_U8_RGB_HWC or FPC_U8_BGR_HWC; a 3-channel ctx.  The bilinear rule
 * restates cv2.resize(INTER_LINEAR) on float32 data; pinned against torch's F.interpolate (same rule), not against
 * OpenCV (absent from this build). */
int fpc_detect_u8_resized(fpc_ctx* ctx, const uint8_t* frames_dev, int n, int src_h, int src_w, int layout);
/* The converted float frames of the last fpc_detect_u8 / fpc_detect_u8_resized call ([n,C,H,W], device) -- for tests. */
const float* fpc_u8_staging(fpc_ctx* ctx);

/* ~ homography_adaptation (python/src/homographies.py:250-324; the caller is
 * InferenceWrapper.run_with_homography_adaptation, inferencewrapper.py:48-68 <- preprocess_coco.py:64-74): the
 * probability maps of n frames aggregated over the un-warped view and `num` perspective views.  For view i the frames
 * are warped by homographies[i] (8 coefficients, the flattened 3x3 with h22 = 1, in the convention of
 * sample_homography :78-182), run through the network, masked, warped back with inverses[i] (NULL: computed here)
 * and accumulated with the validity counts; masks are eroded by an ellipse of radius erosion_radius
 * (config.valid_border_margin, 0 = off); aggregation 0 = 'sum' (count-normalised mean, the reference's default),
 * 1 = 'max'; pixels seen by fewer than num / 3 views become 0.  prob_out_dev [n,H,W] can be fed to fpc_get_points.
 * homographies / inverses are HOST arrays; frames and prob_out are device memory.  Asynchronous.
 * The warps restate torchvision's perspective() and the erosion OpenCV's erode(): both libraries are absent from this
 * build, see csrc/homography.h -- parity for this entry point is unpinned beyond torch's own grid_sample. */
int fpc_homography_adaptation(fpc_ctx* ctx, const float* frames_dev, int n, const float* homographies_host,
                              const float* inverses_host, int num, int erosion_radius, int aggregation,
                              float* prob_out_dev);

/* Runs only the post-processing of fpc_detect on a caller-provided probability map
 * [n,H,W] (device) -- get_points on its own (netutils.py:78-100). Descriptors are
 * sampled from desc_nchw_dev [n,128,H/8,W/8] when it is not NULL. */
int fpc_get_points(fpc_ctx* ctx, const float* prob_map_dev, const float* desc_nchw_dev, int n);

/* ~ get_descriptors(points, descriptors_map, img_h, img_w, settings) on its own (python/src/netutils.py:103-121):
 * bilinear grid_sample (align_corners=True, zero padding) of ONE descriptor map desc_nchw_dev [D,H/8,W/8] at k
 * caller-provided points, then division by the L2 norm (no epsilon: an all-zero sample gives NaN, as there).
 * xy_dev [k][2] float64 (x, y) in pixels of the ctx's H x W frame -- the first two rows of the reference's float64
 * `points`, transposed; the normalisation x / (W / 2) - 1 runs in double and is rounded to float once, as
 * `sample_points.float()` does.  out_dev [k][D] (the reference returns the transpose, [D][k]).  All device memory.
 * Asynchronous on the ctx stream.  Works with descriptor_enabled = 0 too (the map is the caller's). */
int fpc_sample_descriptors(fpc_ctx* ctx, const float* desc_nchw_dev, const double* xy_dev, int k, float* out_dev);

/* --- next row of the path: descriptor matching ----------------
 * ~ cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match(query, train) (python/src/inference.py:88-96):
 * for every query descriptor the nearest train descriptor in L2 (ties: lower index); with
 * cross_check != 0 only mutual nearest neighbours survive; with max_dist > 0 only matches closer
 * than max_dist.  q [nq][128], t [nt][128], match [nq] (train index or -1), dist [nq] (L2
 * distance to the nearest train descriptor; may be NULL) -- all device pointers; nq, nt <=
 * the ctx's keypoint capacity.  Asynchronous on the ctx stream. */
int fpc_match(fpc_ctx* ctx, const float* q_dev, int nq, const float* t_dev, int nt, int cross_check,
              float max_dist, int32_t* match_dev, float* dist_dev);
/* ~ SearchKeyFrameCorrespondence (cpp/src/main.cc:18-29,79-83): for every key-frame descriptor the
 * index of the FIRST current-frame descriptor (in the order given, i.e. descending confidence)
 * whose L2 distance is below `tolerance` (cpp/src/main.cc:54: 0.8), or -1. */
int fpc_first_within(fpc_ctx* ctx, const float* key_dev, int nk, const float* cur_dev, int nc,
                     float tolerance, int32_t* first_dev);

int fpc_results(fpc_ctx* ctx, fpc_device_results* out);
/* Synchronises, then copies the per-frame counts to the host.  FPC_E_NONFINITE (counts delivered all the same) when a
 * frame of the call held a NaN / Inf pixel: "Numerical contract" at the top of this header. */
int fpc_get_counts(fpc_ctx* ctx, int n, int32_t* count_host, int32_t* n_candidates_host);
/* Synchronises; per frame of the last call: the largest logit (fpc_detect and fpc_forward; logits are post-ReLU, >= 0),
 * the largest |value| of the descriptor map (fpc_forward with a desc output only, else 0; a frame's figure may include
 * its neighbour's -- it bounds the tensor), and whether the frame held a non-finite pixel (FPC_F32 only).  Host arrays of
 * n entries, any may be NULL.  What the "Numerical contract" above is checked against. */
int fpc_output_range(fpc_ctx* ctx, int n, float* max_logit_host, float* max_desc_host, int32_t* nonfinite_input_host);
/* Synchronises, then copies frame `frame`'s keypoints: xy [K][2], conf [K],
 * desc [K][128] (desc may be NULL).  Returns K, or FPC_E_CAPACITY if K > cap
 * (nothing is written then; fpc_get_counts gives the size). */
int fpc_get_keypoints(fpc_ctx* ctx, int frame, int cap, int32_t* xy, float* conf, float* desc);

/* Optional per-launch timing for the bench: with `enable`, every kernel launch of
 * fpc_detect / fpc_forward is bracketed by HIP events on the launch stream; records
 * accumulate over calls until the next fpc_set_timing.  enable = n > 1: only every n-th pass over a batch (the first one
 * included; one pass per fpc_detect / fpc_forward / fpc_detect_u8* call, 1 + num per fpc_homography_adaptation) carries
 * the events -- two event records per launch cost 0.7 % of the frame rate at 32 VGA frames per call. */
int fpc_set_timing(fpc_ctx* ctx, int enable);
/* After fpc_sync: number of launches recorded since fpc_set_timing; names[i] (layer) and
 * kernels[i] (kernel symbol, as rocprofv3 prints it) point into ctx-owned storage;
 * ms[i] is the event-to-event duration; flops[i] the ALGORITHMIC FLOPs of that launch
 * (2 x MACs of the direct convolution x its frames, 0 for non-conv kernels); mfma_flops[i]
 * the FLOPs actually issued on the matrix cores (tile / channel padding included;
 * Winograd launches issue 16/36 of their 3x3 convolution's count); bytes[i] the ALGORITHMIC HBM bytes of that launch
 * (its input tensor(s) read once + its output tensor written once, in the mode's storage type, x its frames; weights
 * -- L2-resident -- and data-dependent post-processing traffic not counted).  Any array may be NULL. */
int fpc_get_timings(fpc_ctx* ctx, int cap, const char** names, const char** kernels, float* ms,
                    double* flops, double* mfma_flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* FPC_H */
