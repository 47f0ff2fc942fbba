import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FPC_STREAMS'] = '1'; os.environ['FPC_SPLIT_HEADS'] = '0'; os.environ['FPC_NMS_ASIDE'] = '0'
import fpc_amd
from fpc_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libfpc_diag.so')
from fpc_amd.engine import Engine
import torch
B = 64
for op in sys.argv[1:]:
    os.environ['FPC_STAMP_OP'] = op
    os.environ['FPC_STAMP_FILE'] = '/tmp/fpc_stamps.bin'; os.environ['FPC_STAMP_FULL'] = '1'
    eng = Engine(960, 1280, max_batch=B, dtype='bf16')
    eng.load_state_dict(synth.make_state_dict(0, 7.0))
    frames = torch.from_numpy(synth.make_batch(100, B, 960, 1280)).cuda()
    for _ in range(2):
        eng.detect_async(frames, B)
    eng.sync()
    s = np.fromfile('/tmp/fpc_stamps.bin', dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    w = s[32768:32768 + 4096]
    w = w[w[:, 4] > 0]
    if len(w):
        cyc, us, nt = (w[:, 2] - w[:, 0]).astype(float), (w[:, 3] - w[:, 1]) / 100.0, w[:, 4]
        print(op, 'whole launch per workgroup: %d WGs, tiles %.1f, %.0f cycles per tile, span %.1f us (max %.1f), clock %.2f GHz, launch %.1f us' % (
            len(w), nt.mean(), np.median(cyc / nt), np.median(us), us.max(), np.median(cyc / us) / 1e3, (w[:, 3].max() - w[:, 1].min()) / 100.0))
    s = s[(s[:, 0] != 0) & (s[:, 5] != 0)]
    names = ['first halo chunk', 'phase 1 (all chunks)', 'h -> LDS', 'phase 2', 'epilogue']
    print(op, 'WGs', len(s))
    d = np.diff(s[:, :6], axis=1)
    for i, n in enumerate(names):
        print('  %-28s mean %8.0f  median %8.0f  p90 %8.0f' % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
    print('  tile total mean %.0f median %.0f' % ((s[:, 5] - s[:, 0]).mean(), np.median(s[:, 5] - s[:, 0])))
    rt = (s[:, 7] - s[:, 6]).astype(float)
    ok = rt > 0
    print('  in-kernel clock (median over workgroups): %.2f GHz' % (np.median((s[ok, 5] - s[ok, 0]) / rt[ok]) * 0.1))
    print('  launch span %.0f ; sum of tile totals / span = %.1f concurrent WGs' % (s[:, 5].max() - s[:, 0].min(), (s[:, 5] - s[:, 0]).sum() / (s[:, 5].max() - s[:, 0].min())))
    eng.close()
