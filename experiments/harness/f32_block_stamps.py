import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FPC_STREAMS'] = '1'; os.environ['FPC_SPLIT_HEADS'] = '0'; os.environ['FPC_NMS_ASIDE'] = '0'
import fpc_amd
from fpc_amd import _lib, synth
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libfpc_diag.so')
from fpc_amd.engine import Engine
import torch
B = 32
for op in sys.argv[1:]:
    os.environ['FPC_STAMP_OP'] = op
    os.environ['FPC_STAMP_FILE'] = '/tmp/fpc_stamps.bin'
    eng = Engine(480, 640, max_batch=B)
    eng.load_state_dict(synth.make_state_dict(0, 7.0))
    frames = torch.from_numpy(synth.make_batch(100, B)).cuda()
    for _ in range(2):
        eng.detect_async(frames, B)
    eng.sync()
    s = np.fromfile('/tmp/fpc_stamps.bin', dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    s = s[(s[:, 0] != 0) & (s[:, 5] != 0)]
    print(op, 'WGs', len(s))
    d = np.diff(s[:, :6], axis=1)
    for i in range(5):
        print('  stamp %d->%d  mean %8.0f  median %8.0f' % (i, i + 1, d[:, i].mean(), np.median(d[:, i])))
    print('  tile total mean %.0f' % (s[:, 5] - s[:, 0]).mean())
    rt = (s[:, 7] - s[:, 6]).astype(float)      # 100 MHz ticks between stamps 0 and 5 (wblock36_kernel records them)
    ok = rt > 0
    if ok.any():
        print('  in-kernel clock (median over workgroups): %.2f GHz' % (np.median((s[ok, 5] - s[ok, 0]) / rt[ok]) * 0.1))
    eng.close()
