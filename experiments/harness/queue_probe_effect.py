"""How much a context's frame rate depends on WHICH hardware queues its streams got:
    for k in 0 1 2 3 4 5; do python experiments/harness/queue_probe_effect.py $k; done                      # with fpc_create's probe
    for k in 0 1 2 3 4 5; do FPC_QUEUE_PROBE=0 python experiments/harness/queue_probe_effect.py $k; done    # without
k = unrelated streams the process makes (and uses once) before the engine; the workload is bench.py's bounded pass of
hd64-bf16 (64 x 1280x960, bf16).  Round 4, one MI355X, without the probe: 12 790 / 12 770 / 12 110 / 11 620 / 12 780 / 12 730
frames/s for k = 0 .. 5; with it 12 780-12 790 for every k."""
import sys, json, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from fpc_amd import synth
k = int(sys.argv[1])
torch.cuda.set_device(0)
hold = [torch.cuda.Stream() for _ in range(k)]      # dummy streams created before the engine's
for s in hold:
    with torch.cuda.stream(s):
        torch.zeros(1, device='cuda')
torch.cuda.synchronize()
sd = synth.make_state_dict(0, dustbin_bias=7.0)
r = bench.side_workload('hd64-bf16', sd, 0)
print('dummy streams', k, 'value', r['value'], r['ms_per_step'])
