#!/usr/bin/env python3
"""Fixtures F7: outputs of the REFERENCE's own C++ network (superpoint::SPModel, cpp/src/model.cc) on seeded
parameters and frames.  The network is the reference's source compiled unmodified against this image's
libtorch (oracle/Makefile.ref -> oracle/_ref/ref_vgg_forward); this script only writes its input files,
runs it, and stores what it returns.  Run in the build container (needs /root/reference):

    make -C oracle -f Makefile.ref && python tests/golden/make_golden_vgg.py

  f7_vgg_32x48.npz  : logits [1,65,4,6] and desc [1,256,4,6] in full, for a 32x48 frame
  f7_vgg_qvga.npz   : strided probes of both maps for a 240x320 frame + checksums
  f7_vgg_names.txt  : SPModel's named_parameters() (name + shape) as the binary prints them
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import fpc_amd  # noqa: E402,F401
from fpc_amd import arch, synth  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "ref_vgg_forward")


def run_reference(sd, frames, repeats=None):
    """frames float32 [n,1,h,w] -> (point [n,65,hc,wc], desc [n,256,hc,wc], names text); with `repeats` the names
    text is replaced by the binary's stderr (mean forward seconds and thread count)."""
    with tempfile.TemporaryDirectory() as td:
        pf, inf, outf = (os.path.join(td, x) for x in ("params.bin", "input.bin", "output.bin"))
        with open(pf, "wb") as f:
            f.write(struct.pack("<i", len(sd)))
            for name, v in sd.items():
                nb = name.encode()
                f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", v.ndim))
                f.write(struct.pack("<%dq" % v.ndim, *v.shape))
                f.write(np.ascontiguousarray(v, np.float32).tobytes())
        n, _, h, w = frames.shape
        with open(inf, "wb") as f:
            f.write(struct.pack("<iii", n, h, w) + np.ascontiguousarray(frames, np.float32).tobytes())
        cp = subprocess.run([BIN, pf, inf, outf] + ([str(repeats)] if repeats is not None else []), check=True,
                            capture_output=True, text=True)
        names = cp.stdout if repeats is None else cp.stderr
        raw = open(outf, "rb").read()
    n2, hc, wc = struct.unpack("<iii", raw[:12])
    a = np.frombuffer(raw[12:], np.float32)
    point = a[:n2 * 65 * hc * wc].reshape(n2, 65, hc, wc).copy()
    desc = a[n2 * 65 * hc * wc:].reshape(n2, 256, hc, wc).copy()
    return point, desc, names


def gray(seed, h, w):
    return synth.make_batch(seed, 1, h, w, gray=True)[:, :1].copy()


if __name__ == "__main__":
    sd = synth.make_vgg_state_dict(31, 3.0)
    fr = gray(400, 32, 48)
    point, desc, names = run_reference(sd, fr)
    spec = arch.vgg_state_dict_spec()
    got = [ln.split() for ln in names.strip().splitlines()]
    assert [g[0] for g in got] == list(spec.keys()), "arch.vgg_state_dict_spec() != SPModel::named_parameters()"
    assert all(tuple(int(x) for x in g[1:]) == tuple(spec[g[0]]) for g in got)
    open(os.path.join(HERE, "f7_vgg_names.txt"), "w").write(names)
    np.savez(os.path.join(HERE, "f7_vgg_32x48.npz"), seed_weights=31, dustbin_bias=3.0, seed_frame=400,
             logits=point, desc=desc)
    print("F7 32x48", point.shape, desc.shape, "desc norm", float(np.linalg.norm(desc[0, :, 1, 2])))
    sd = synth.make_vgg_state_dict(32, 3.0)
    fr = gray(401, 240, 320)
    point, desc, _ = run_reference(sd, fr)
    np.savez(os.path.join(HERE, "f7_vgg_qvga.npz"), seed_weights=32, dustbin_bias=3.0, seed_frame=401, h=240, w=320,
             logits_probe=point.ravel()[::7].copy(), desc_probe=desc.ravel()[::11].copy(),
             logits_sum=float(point.astype(np.float64).sum()), desc_abs_sum=float(np.abs(desc).astype(np.float64).sum()))
    print("F7 qvga", point.shape, "logit range", float(point.min()), float(point.max()))
