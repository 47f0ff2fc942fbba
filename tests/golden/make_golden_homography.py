#!/usr/bin/env python3
"""Fixture F10: homographies drawn by the REFERENCE's own `sample_homography`
(/root/reference/python/src/homographies.py:78-192), run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_homography.py

The module itself cannot be imported here (its first lines import torchvision and cv2, both absent), but the three
functions this fixture needs -- `truncated_normal` :64-67, `random_uniform` :70-75, `sample_homography` :78-192 -- use
torch, scipy and `pi` only.  This script reads the module's text, lets `ast` pick exactly those three function
definitions, compiles THEM (nothing of the file is stored or re-typed here) in a namespace that holds the names they
use, and records what they return.  No stand-in for a missing library is written: nothing of torchvision / cv2 is
touched by these functions.

Stored (data only): for three configurations -- the class defaults (:33-49), `init_for_preprocess` (:51-61, what
preprocess_coco.py runs) and perspective alone -- 1000 homographies of a 480 x 640 frame each, float32 [1000, 8], with
the seed that torch's and numpy's global generators were given.  The reference's stream cannot be reproduced by the
build's sampler (numpy Generator), so the test compares DISTRIBUTIONS (tests/test_abi_and_host.py).
"""
import ast
import os
import sys
from math import pi

import numpy as np
import torch
from scipy.stats import truncnorm

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/python/src/homographies.py"
WANTED = ("truncated_normal", "random_uniform", "sample_homography")


def reference_functions():
    text = open(SRC).read()
    tree = ast.parse(text)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in picked) == sorted(WANTED)
    ns = {"torch": torch, "np": np, "truncnorm": truncnorm, "pi": pi}
    exec(compile(ast.Module(body=picked, type_ignores=[]), SRC, "exec"), ns)
    return ns


def draw(fn, n, seed, **kw):
    torch.manual_seed(seed)
    np.random.seed(seed)        # scipy's rvs() without random_state draws from numpy's global generator
    return np.stack([fn((480, 640), **kw).numpy().astype(np.float32) for _ in range(n)])


def main():
    ns = reference_functions()
    fn = ns["sample_homography"]
    n, seed = 1000, 20260
    defaults = draw(fn, n, seed)
    # HomographyConfig.init_for_preprocess (:51-61) as preprocess_coco.py passes it on (homographies.py:272-282)
    pre = draw(fn, n, seed + 1, perspective=True, scaling=True, rotation=True, translation=True, n_scales=5, n_angles=25,
               scaling_amplitude=0.2, perspective_amplitude_x=0.2, perspective_amplitude_y=0.2, patch_ratio=0.85,
               max_angle=pi / 2, allow_artifacts=True, translation_overflow=0.)
    persp_only = draw(fn, 50, seed + 2, scaling=False, rotation=False, translation=False)
    no_scaling = draw(fn, n, seed + 3, scaling=False)
    out = os.path.join(HERE, "f10_sample_homography.npz")
    np.savez_compressed(out, defaults=defaults, preprocess=pre, perspective_only=persp_only, no_scaling=no_scaling,
                        seed=np.int64(seed), shape=np.array([480, 640]))
    print("wrote", out, {k: v.shape for k, v in (("defaults", defaults), ("preprocess", pre))})
    print("max |h7|, |h8| (defaults):", np.abs(defaults[:, 6:]).max(0), "(preprocess):", np.abs(pre[:, 6:]).max(0))
    print("perspective only:", persp_only[:2])


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()
