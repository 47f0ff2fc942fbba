"""Round-3 parity tests (pytest -m gpu).

The bf16 stem was rebuilt in round 3 (csrc/stem_bf16.h: another tile shape, another lane -> pixel map, weights in
registers, other LDS layouts).  Its arithmetic is meant to be round 2's to the bit -- same fragments, same K order,
same roundings -- so besides the oracle tests of tests/test_gpu_parity.py (which hold it to the bf16-emulating oracle
like every layer) the stand-alone harness runs it beside round 2's kernel and the wave-specialised variant
(experiments/) on the same random frames and compares the pooled maps BIT FOR BIT: at a size smaller than a tile row,
at an odd geometry whose last tiles hang over two edges, for the gray instance, and at HD."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "experiments", "harness", "stem_bf16_bench.hip")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("cin", [3, 1])
def test_bf16_stem_is_bit_identical_to_round_2s_kernel(cin, tmp_path):
    exe = str(tmp_path / ("stem_bench_c%d" % cin))
    subprocess.run([_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-w", "-DSTEM_CIN=%d" % cin, "-o", exe, HARNESS],
                   check=True, timeout=600)
    # frames, H, W, grid of the two-per-CU kernels (a multiple of 8), grid of the one-per-CU variant
    cases = [(2, 64, 96, 64, 8), (3, 200, 264, 64, 32), (1, 16, 24, 8, 8), (2, 960, 1280, 512, 256)]
    for b, h, w, g, g3 in cases:
        r = subprocess.run([exe, str(b), str(h), str(w), str(g), str(g3)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=120)
        out = r.stdout.decode()
        assert r.returncode == 0, out
        assert "compare: stem_bf16 0, stem_bf16_ws 0 of" in out, out
