"""The network of the path restated with torch.nn.functional on the CPU -- TEST / BASELINE INFRASTRUCTURE ONLY.

The reference's CPU path IS eager PyTorch on the host cores (python/src/superpoint.py:91-115 run with
settings.cuda = False; oneDNN convolutions, unfused BatchNorm / ReLU / add).  The reference itself cannot travel to the
GPU box, so bench.py's `cpu_baseline` times this restatement there: the same operators in the same order, written
from the module definitions (superpoint.py:8-61, resnet_blocks.py:4-41), not copied from them.  Checked against the
reference's own outputs (fixture F5) in tests/test_oracle_vs_golden.py.  Like everything under oracle/, only tests/,
smoke() and bench.py's cpu_baseline leg may import it; the product never does.
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, EPS)


def _block(x, sd, p, stride, proj):
    """ResNetBlock.forward (resnet_blocks.py:14-27)."""
    h = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    y = _bn(F.conv2d(h, sd[p + ".conv2.weight"]), sd, p + ".bn2")
    idt = x
    if proj:
        idt = _bn(F.conv2d(x, sd[p + ".identity_downsample.0.weight"], None, stride), sd, p + ".identity_downsample.1")
    return F.relu(y + idt)


def _stage(x, sd, p, stride):
    return _block(_block(x, sd, p + ".0", stride, True), sd, p + ".1", 1, False)


def to_torch(state_dict):
    return {k: torch.from_numpy(v.copy()) for k, v in state_dict.items() if v.dtype.kind == "f"}


@torch.no_grad()
def forward(image, sd, descriptor_enabled=True):
    """image [B,3,H,W] float tensor, sd = to_torch(state_dict) -> (prob_map [B,H,W], desc [B,128,H/8,W/8], logits)."""
    x = F.relu(_bn(F.conv2d(image, sd["encoder.conv1.weight"], None, 2, 3), sd, "encoder.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    x = _stage(x, sd, "encoder.layer1", 1)
    feat = _stage(x, sd, "encoder.layer2", 2)
    logits = _stage(feat, sd, "detector.layer", 1)
    if descriptor_enabled:
        y = _stage(feat, sd, "descriptor.layer_in", 2)
        y = F.conv_transpose2d(y, sd["descriptor.up_sample.weight"], sd["descriptor.up_sample.bias"], 2, 1, 1)
        y = F.relu(_bn(y, sd, "descriptor.bn"))
        desc = _stage(torch.cat([y, feat], 1), sd, "descriptor.layer_out", 1)
    else:
        desc = torch.zeros((image.shape[0], 128, image.shape[2] // 8, image.shape[3] // 8))
    e = torch.exp(logits)
    sm = e / (e.sum(1, keepdim=True) + .00001)                  # superpoint.py:111-112
    b, _, hc, wc = sm.shape
    prob = sm[:, :-1].permute(0, 2, 3, 1).reshape(b, hc, wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, hc * 8, wc * 8)
    return prob, desc, logits
