#!/usr/bin/env python3
"""Headline benchmark: VGA SuperPoint forward + NMS + descriptors, frames/s.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (fpc_detect: network, exp-softmax,
depth-to-space, threshold, greedy NMS, sort, border crop, descriptor sampling) over
one batch of 32 synthetic 640x480 frames that is already resident in HBM
(BASELINE.json configs[1]).  With N ranks every rank runs its own 32-frame batch
(weak scaling; configs[2] = 256 frames over 8 GPUs); the only collective is the
start-up broadcast of the packed weights.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fpc_amd  # noqa: E402,F401
from fpc_amd import arch, dist as fdist, synth  # noqa: E402
from fpc_amd.engine import Engine  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 matrix peak (dense, f32 in / f32 acc)
PEAK_BF16_MFMA_TFLOPS = 2516.8  # ibid.: BF16 MFMA dense = 16x the FP32 matrix rate (~2.5 PF)
PEAK_TFLOPS = PEAK_F32_MFMA_TFLOPS     # ceiling for ALGORITHMIC flops in the selected arithmetic mode
PEAK_ISSUED_TFLOPS = PEAK_F32_MFMA_TFLOPS  # ceiling for the MFMA instructions really issued
BATCH = 32
# HBM bytes per FRAME of each kernel symbol from the rocprofv3 PMC passes in
# profiles/r01j_pmc_summary_serial.csv (one stream, default plan): (2*FETCH_SIZE + WRITE_SIZE) KiB per
# dispatch / 32 frames, averaged over the layers that share the symbol (the x2 on FETCH_SIZE is the gfx950
# correction of MI355X_MICROARCH.md for 16 B/lane reads; FETCH_SIZE and WRITE_SIZE in separate passes).
TRAFFIC_BYTES_PER_FRAME = {
    "wblock_mfma_kernel<32, 4, 128>": 254242720 / 32.0,
    "wblock_mfma_kernel<32, 2, 64>": 326846528 / 32.0,
    "wblock_mfma_kernel<32, 3, 72>": 174261248 / 32.0,
    "stem_pool_kernel": 315067808 / 32.0,
    # split-operand kernels (profiles/r01j_pmc_summary_serial_f32_split_f16.csv; the bf16-term twins move the same bytes)
    "block_h2_kernel<6, 20, 1, 3, 64, 2, 2, 2, 2, 128>": 307176736 / 32.0,
    "block_h2_kernel<8, 16, 1, 3, 64, 2, 2, 2, 1, 64>": 460159776 / 32.0,
    "block_x3_kernel<6, 20, 1, 3, 64, 2, 2, 2, 2, 128>": 307176736 / 32.0,
    "block_x3_kernel<8, 16, 1, 3, 64, 2, 2, 2, 1, 64>": 460159776 / 32.0,
    "stem_pool_x3_kernel": 315311808 / 32.0,
}
H, W = 480, 640


def cpu_baseline(state_dict, frames):
    """The CPU oracle (oracle/fpc_oracle.c, kind "port") timed on this host's cores on a
    bounded sample of the same workload.  Rank 0, N=1 only."""
    from oracle import oracle
    spec = arch.state_dict_spec()
    n = frames.shape[0]
    threads = oracle.max_threads()
    oracle.forward(frames[:1], state_dict, spec)          # warm-up
    t0 = time.perf_counter()
    kept = 0
    for i in range(n):
        prob, desc, _ = oracle.forward(frames[i:i + 1], state_dict, spec)
        xs, ys, _, _ = oracle.get_points(prob[0])
        oracle.get_descriptors(desc[0], xs, ys, H, W)
        kept += len(xs)
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d of the bench's %dx%d frames, full path (forward + get_points + get_descriptors), "
                      "oracle/fpc_oracle.c with %d OpenMP threads, %.1f s" % (n, W, H, threads, dt)}


def cpu_baseline_vgg_reference(sd, frames):
    """kind "reference": the reference's own cpp/src/model.cc (oracle/_ref/ref_vgg_forward, built by
    oracle/Makefile.ref) timed on this host's cores -- the forward pass only, which is all of cpp/ that can be built."""
    import importlib.util
    path = os.path.join(ROOT, "tests", "golden", "make_golden_vgg.py")
    spec = importlib.util.spec_from_file_location("make_golden_vgg", path)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    if not os.path.exists(mg.BIN):
        return None
    n, reps = frames.shape[0], 3
    _, _, err = mg.run_reference(sd, frames, repeats=reps)
    kv = dict(ln.split() for ln in err.strip().splitlines() if len(ln.split()) == 2)
    sec = float(kv["forward_seconds"])
    return {"value": round(n / sec, 3), "unit": "frames/s", "cores": int(kv.get("threads", 0)), "kind": "reference",
            "sample": "%d of the bench's %dx%d gray frames, SPModel::forward only (cpp/src/model.cc compiled unmodified "
                      "with libtorch CPU, %d runs of %.2f s); the rest of cpp/ needs TRTorch/OpenCV" % (n, W, H, reps, sec)}


def host_fed_rates(sd, frames_np, local, dtype, steps=12):
    """H2D-inclusive rates (the second figure, never `value`): every step uploads its batch from
    pinned host memory and runs the path; two contexts on two streams so that batch k+1 uploads while batch k
    computes.  (a) the reference's input, fp32 RGB [n,3,H,W]; (b) 8-bit RGB HWC frames converted on the device
    (fpc_detect_u8)."""
    dev = torch.device("cuda", local)
    res = {}
    u8 = np.clip(np.rint(frames_np.transpose(0, 2, 3, 1) * 255.0), 0, 255).astype(np.uint8)
    for tag, host in (("f32_rgb_nchw", torch.from_numpy(frames_np).pin_memory()),
                      ("u8_rgb_hwc", torch.from_numpy(np.ascontiguousarray(u8)).pin_memory())):
        ctxs = []
        for _ in range(2):
            st = torch.cuda.Stream(device=dev)
            e = Engine(H, W, max_batch=BATCH, device=local, dtype=dtype)
            e.load_state_dict(sd)
            with torch.cuda.stream(st):
                e.use_torch_stream()
            ctxs.append((st, e, torch.empty_like(host, device=dev)))

        def step(i):
            st, e, buf = ctxs[i % 2]
            with torch.cuda.stream(st):
                buf.copy_(host, non_blocking=True)
                if tag == "u8_rgb_hwc":
                    e.detect_u8_async(buf, BATCH, "rgb_hwc")
                else:
                    e.detect_async(buf, BATCH)
        for i in range(4):
            step(i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        res[tag] = {"value": round(BATCH * steps / dt, 2), "unit": "frames/s", "steps": steps,
                    "h2d_bytes_per_frame": int(host[0].numel() * host.element_size()),
                    "h2d_gbytes_per_s": round(host.numel() * host.element_size() * steps / dt / 1e9, 2)}
        for _, e, _b in ctxs:
            e.close()
    return res


def symbol_stats(timings, steps):
    """kernel symbol -> dict(avg ms, algorithmic / MFMA-issued FLOPs per launch, launches per step, layers)."""
    by = {}
    layers = {}
    for name, kern, ms, fl, mf in timings:
        by.setdefault(kern, []).append((ms, fl, mf))
        layers.setdefault(kern, set()).add(name)
    out = {}
    for k, v in by.items():
        ms = float(np.mean([m for m, _, _ in v]))
        out[k] = {"avg_launch_ms": ms, "flops": float(np.mean([f for _, f, _ in v])),
                  "mfma_flops": float(np.mean([f for _, _, f in v])), "launches_per_step": len(v) // max(1, steps),
                  "layers": len(layers[k]), "total_ms": float(np.sum([m for m, _, _ in v])) / max(1, steps)}
    return out


def roofline_entry(sym, st, step_ms):
    ach = st["flops"] / (st["avg_launch_ms"] * 1e-3) / 1e12
    frames = BATCH * st["layers"] / max(1, st["launches_per_step"])
    return {"bound": "mfma", "kernel": sym, "achieved": round(ach, 3), "peak": PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS, 4),
            "traffic": (round(TRAFFIC_BYTES_PER_FRAME[sym] * frames)
                        if sym in TRAFFIC_BYTES_PER_FRAME and (H, W) == (480, 640) else None),
            "avg_launch_ms": round(st["avg_launch_ms"], 4), "frames_per_launch": frames,
            "flops_per_launch": st["flops"], "mfma_issued_flops_per_launch": st["mfma_flops"],
            "mfma_issued_frac": round(st["mfma_flops"] / (st["avg_launch_ms"] * 1e-3) / 1e12 / PEAK_ISSUED_TFLOPS, 4),
            "launches_per_step": st["launches_per_step"], "share_of_step": round(st["total_ms"] / step_ms, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("FPC_BENCH_CONTEXTS", "1")),
                    help="fpc contexts used round-robin (double buffering: batch k+1 starts while batch k drains)")
    ap.add_argument("--gray", action="store_true",
                    help="feed gray frames [n,1,H,W] (in_channels=1: stem filters summed over the input channels); "
                         "default is the reference network's 3-channel input")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra one-stream pass that gives clean per-kernel durations")
    ap.add_argument("--no-timing-events", action="store_true",
                    help="do not bracket launches with HIP events (roofline then comes from a separate pass)")
    ap.add_argument("--workload", choices=["vga32", "hd64-bf16", "qvga32-magicpoint"], default="vga32",
                    help="vga32 = BASELINE.json configs[1] (the headline; default); hd64-bf16 = configs[4]: 64 frames of "
                         "1280x960 per step, bf16 activations / weights with fp32 accumulation")
    ap.add_argument("--dtype", choices=["f32", "f32_split", "f32_split_f16"], default="f32",
                    help="vga32 only. f32 (default): fp32 MFMA (v_mfma_f32_32x32x2_f32, Winograd where it pays). "
                         "f32_split: fp32 tensors, every product as six bf16 MFMAs on exactly split operands "
                         "(block_x3.h) -- same 1e-4 parity bar, reported beside the headline as `split_operand_mode`. "
                         "f32_split_f16: the same with two fp16 terms per operand and three MFMAs per product")
    ap.add_argument("--arch", choices=["resnet", "vgg"], default="resnet",
                    help="resnet (default): the Python network of BASELINE.json's configs. vgg: the cpp/ frontend's "
                         "superpoint::SPModel (SURVEY 8f rank 4), gray frames, 52 GFLOP per VGA frame")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-fed (H2D-inclusive) passes")
    ap.add_argument("--no-alt-pass", action="store_true", help="skip the extra pass in the other fp32 arithmetic mode")
    args = ap.parse_args()
    global H, W, BATCH, PEAK_TFLOPS, PEAK_ISSUED_TFLOPS
    dtype = args.dtype
    if dtype in ("f32_split", "f32_split_f16"):
        PEAK_TFLOPS = PEAK_BF16_MFMA_TFLOPS / (6.0 if dtype == "f32_split" else 3.0)   # six bf16 / three fp16 MFMAs per product
        PEAK_ISSUED_TFLOPS = PEAK_BF16_MFMA_TFLOPS
    magic = args.workload == "qvga32-magicpoint"   # configs[3]: detector only, 240x320, NMS r = 4
    if magic:
        H, W = 240, 320
        args.no_host_fed = True
    if args.workload == "hd64-bf16":
        H, W, BATCH, dtype, PEAK_TFLOPS = 960, 1280, 64, "bf16", PEAK_BF16_MFMA_TFLOPS
        PEAK_ISSUED_TFLOPS = PEAK_BF16_MFMA_TFLOPS

    rank, world, local = fdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    local = local % max(1, torch.cuda.device_count())   # (a gloo rehearsal may put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # synthetic checkpoint in the reference's layout; rank 0 packs and broadcasts it
    vgg = args.arch == "vgg"
    if vgg:
        args.gray, args.no_host_fed, args.no_alt_pass = True, True, True
    sd = (synth.make_vgg_state_dict(0, dustbin_bias=5.5) if vgg else synth.make_state_dict(0, dustbin_bias=7.0)) if rank == 0 else None
    cin = 1 if args.gray else 3
    eng = Engine(H, W, max_batch=BATCH, device=local, in_channels=cin, dtype=dtype, arch=args.arch, descriptor_enabled=not magic)
    fdist.broadcast_packed_weights(eng, sd)
    engs = [eng]
    for _ in range(1, max(1, args.contexts)):
        e2 = Engine(H, W, max_batch=BATCH, device=local, in_channels=cin, dtype=dtype, arch=args.arch, descriptor_enabled=not magic)
        e2.import_packed(eng.export_packed())
        engs.append(e2)
    # this rank's frames: seeds 100 + 32*rank ... (configs[2]: seeds 100..355 over 8 GPUs)
    frames_np = synth.make_batch(100 + BATCH * rank, BATCH, H, W, gray=args.gray)
    if args.gray:
        frames_np = np.ascontiguousarray(frames_np[:, :1])
    frames = torch.from_numpy(frames_np).to(dev)
    torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        engs[i % len(engs)].detect_async(frames, BATCH)
    for e_ in engs:
        e_.sync()
    cnt, ncand = eng.counts(BATCH)

    use_events = not args.no_timing_events
    for e_ in engs:
        e_.set_timing(use_events)
    per_kernel, per_symbol = {}, {}
    fdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        engs[i % len(engs)].detect_async(frames, BATCH)
    for e_ in engs:
        e_.sync()
    torch.cuda.synchronize(dev)
    fdist.barrier()
    dt = time.perf_counter() - t0
    if use_events:
        # every launch of the timed region: layer name / kernel symbol -> [(ms, algorithmic flops, mfma flops)]
        for e_ in engs:
            for name, kern, ms, fl, mf in e_.timings():
                per_kernel.setdefault(name, []).append((ms, fl, mf))
    timed_timings = [t for e_ in engs for t in e_.timings()] if use_events else []
    for e_ in engs:
        e_.set_timing(False)
    dt = fdist.max_over_ranks(dt)
    serial = None
    if use_events and rank == 0 and world == 1 and not args.no_serial_pass:
        # same kernels, one stream: clean per-kernel durations (not part of `value`)
        os.environ["FPC_STREAMS"], os.environ["FPC_SPLIT_HEADS"], os.environ["FPC_NMS_ASIDE"] = "1", "0", "0"
        e1 = Engine(H, W, max_batch=BATCH, device=local, in_channels=cin, dtype=dtype, arch=args.arch, descriptor_enabled=not magic)
        e1.import_packed(eng.export_packed())
        for _ in range(2):
            e1.detect_async(frames, BATCH)
        e1.sync()
        e1.set_timing(True)
        ks = min(10, args.steps)
        t1 = time.perf_counter()
        for _ in range(ks):
            e1.detect_async(frames, BATCH)
        e1.sync()
        d1 = time.perf_counter() - t1
        serial = (symbol_stats(e1.timings(), ks), d1 / ks * 1e3, e1.timings(), ks)
        e1.close()
    alt = {}
    if rank == 0 and world == 1 and not args.no_alt_pass and args.workload == "vga32" and not vgg:
        # the other fp32 arithmetic modes on the same frames (not part of `value`)
        os.environ.pop("FPC_STREAMS", None)
        os.environ.pop("FPC_SPLIT_HEADS", None)
        os.environ.pop("FPC_NMS_ASIDE", None)
        for adt in ("f32", "f32_split", "f32_split_f16"):
            if adt == dtype:
                continue
            ea = Engine(H, W, max_batch=BATCH, device=local, in_channels=cin, dtype=adt)
            ea.load_state_dict(sd)
            for _ in range(3):
                ea.detect_async(frames, BATCH)
            ea.sync()
            ks = min(20, max(5, args.steps))
            t1 = time.perf_counter()
            for _ in range(ks):
                ea.detect_async(frames, BATCH)
            ea.sync()
            d1 = time.perf_counter() - t1
            acnt, _ = ea.counts(BATCH)
            key = {"f32": "plain_f32_mfma_mode", "f32_split": "split_operand_mode", "f32_split_f16": "split_operand_fp16_mode"}[adt]
            alt[key] = {"dtype": adt, "value": round(BATCH * ks / d1, 2), "unit": "frames/s", "steps": ks,
                        "ms_per_step": round(d1 / ks * 1e3, 4), "keypoints_per_frame": round(float(np.mean(acnt)), 1),
                        "same_keypoint_counts_as_headline_mode": bool(np.array_equal(acnt, cnt))}
            ea.close()
    host_fed = None
    if rank == 0 and world == 1 and not args.no_host_fed and args.workload == "vga32" and not args.gray:
        host_fed = host_fed_rates(sd, frames_np, local, dtype)
    total_frames = BATCH * args.steps * world

    if rank == 0:
        value = total_frames / dt
        flops_frame = 2.0 * (arch.vgg_conv_macs(H, W) if vgg else arch.conv_macs(H, W, descriptor=not magic))
        wl = ("batch=32 640x480 frames per GPU, super_point checkpoint layout, fp32 "
              "(BASELINE.json configs[1]; configs[2] when n_gpus=8)" +
              ("; products as six bf16 MFMAs on exactly split fp32 operands" if dtype == "f32_split" else
               "; products as three fp16 MFMAs on two-term split fp32 operands" if dtype == "f32_split_f16" else "")
              ) if dtype != "bf16" else (
              "batch=64 1280x960 frames per GPU, super_point checkpoint layout, bf16 activations/weights, fp32 "
              "accumulation, fp32 post-processing (BASELINE.json configs[4])")
        if magic:
            wl = "batch=32 240x320 frames per GPU, detector only (MagicPoint), NMS r=4, " + dtype + " (BASELINE.json configs[3])"
        if vgg:
            wl = "batch=32 640x480 gray frames per GPU, the cpp/ frontend's network (superpoint::SPModel, 256-D), " + dtype
        out = {
            "metric": "frames/sec (%s) SuperPoint fwd+NMS+descriptors" % ("QVGA 320x240, detector only" if magic else "VGA 640x480" if dtype != "bf16" else "HD 1280x960"),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16", "f32_split": "f32 (3 x bf16 split operands, 6 MFMA per product, f32 accumulate)",
                      "f32_split_f16": "f32 (2 x fp16 split operands, 3 MFMA per product, f32 accumulate)"}[dtype],
            "data": "synthetic (seeded frames + seeded checkpoint in the reference's layout)",
            "config": {"workload": wl, "frames_per_step_per_gpu": BATCH, "height": H, "width": W, "input_channels": cin,
                       "parallelism": "frame-sharded x%d, no data-path collective" % world},
            "frames_per_sec_per_gpu": round(value / world, 2),
            "whole_path_tflops": round(value * flops_frame / 1e12, 3),
            "whole_path_frac_of_%s_mfma_peak" % ("f32" if dtype == "f32" else "bf16"): round(
                value / world * flops_frame * {"f32_split": 6.0, "f32_split_f16": 3.0}.get(dtype, 1.0) / 1e12 /
                (PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS), 4),
            "keypoints_per_frame": round(float(np.mean(cnt)), 1),
            "candidates_per_frame": round(float(np.mean(ncand)), 1),
        }
        if per_kernel:
            # dominant kernel = the kernel SYMBOL (as rocprofv3 aggregates) with the largest share of
            # the step; achieved = its ALGORITHMIC FLOPs per launch / its average launch duration, from
            # HIP events recorded on the launch streams inside the timed region.  Winograd launches issue
            # fewer FLOPs on the matrix cores than the direct convolution they compute (16/36 of the 3x3),
            # so `achieved` counts work the matrix cores did not have to do; `mfma_issued_frac` is their
            # utilisation by what was really issued.  In the timed region up to four streams run
            # concurrently, so a launch's duration includes the time it shares the GPU with others;
            # `roofline_serial` repeats the measurement with one stream (kernels alone on the GPU).
            stats = symbol_stats(timed_timings, args.steps)
            sym = max((k for k in stats if stats[k]["flops"] > 0), key=lambda k: stats[k]["total_ms"])
            out["roofline"] = roofline_entry(sym, stats[sym], dt / args.steps * 1e3)
            sum_ms = {k: float(np.sum([m for m, _, _ in v])) / args.steps for k, v in per_kernel.items()}
            out["layer_ms_per_step_concurrent"] = {k: round(v, 4) for k, v in sum_ms.items()}
            out["mfma_issued_tflops_whole_step"] = round(
                sum(f for v in per_kernel.values() for _, _, f in v) / args.steps / (dt / args.steps) / 1e12, 3)
            if serial is not None:
                sstats, sms, stim, ks = serial
                out["roofline_serial"] = roofline_entry(sym, sstats[sym], sms)
                out["roofline_serial"]["ms_per_step_one_stream"] = round(sms, 4)
                lay = {}
                for name, kern, ms, fl, mf in stim:
                    lay.setdefault(name, []).append(ms)
                out["layer_ms_serial"] = {k: round(float(np.mean(v)), 4) for k, v in lay.items()}
        if host_fed is not None:
            out["host_fed"] = host_fed
        out.update(alt)
        if world == 1 and not args.no_cpu_baseline and vgg:
            cb = cpu_baseline_vgg_reference(sd, frames_np[:4])
            if cb is not None:
                out["cpu_baseline"] = cb
        elif world == 1 and not args.no_cpu_baseline:
            ncb = 8 if H * W <= 480 * 640 else 2
            cb_frames = frames_np[:ncb] if not args.gray else np.repeat(frames_np[:ncb], 3, axis=1)
            out["cpu_baseline"] = cpu_baseline(sd, cb_frames)
        print(json.dumps(out))
    for e_ in engs:
        e_.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
