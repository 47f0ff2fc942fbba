"""Static description of the SuperPoint / MagicPoint network of the reference
(python/src/superpoint.py:8-61, python/src/resnet_blocks.py:4-41) as a table of
checkpoint entries.  Nothing here computes; the table is what the loader
(`weights.py`), the synthetic checkpoint generator (`synth.py`) and the tests
agree on.

Key names and shapes are those of ``ckpt['model_state_dict']`` as the reference
trainer writes it (python/src/saveutils.py:54-63); SURVEY.md table W.
"""
from collections import OrderedDict

BN_EPS = 1e-5          # torch.nn.BatchNorm2d default (resnet_blocks.py:8,10,35)
CELL = 8               # settings.py:7
DESC_DIM = 128         # superpoint.py:49-50
DET_CH = 65            # superpoint.py:32  (64 cell positions + dustbin)


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def _block(prefix, cin, cout, proj):
    """One ResNetBlock (resnet_blocks.py:5-12); `proj` = has identity_downsample."""
    e = [(prefix + ".conv1.weight", (cout, cin, 3, 3))]
    e += _bn(prefix + ".bn1", cout)
    e += [(prefix + ".conv2.weight", (cout, cout, 1, 1))]
    e += _bn(prefix + ".bn2", cout)
    if proj:
        e += [(prefix + ".identity_downsample.0.weight", (cout, cin, 1, 1))]
        e += _bn(prefix + ".identity_downsample.1", cout)
    return e


def _stage(prefix, cin, cout):
    """make_resnet_layers(2, cin, cout, stride) (resnet_blocks.py:30-41): the first
    block always carries a projection shortcut, the second never does."""
    return _block(prefix + ".0", cin, cout, True) + _block(prefix + ".1", cout, cout, False)


def state_dict_spec():
    """Ordered {name: shape} of all 163 checkpoint entries."""
    e = [("encoder.conv1.weight", (64, 3, 7, 7))]
    e += _bn("encoder.bn1", 64)
    e += _stage("encoder.layer1", 64, 64)
    e += _stage("encoder.layer2", 64, 128)
    e += _stage("detector.layer", 128, DET_CH)
    e += _stage("descriptor.layer_in", 128, 256)
    # ConvTranspose2d weight is (Cin, Cout, kH, kW) (superpoint.py:45)
    e += [("descriptor.up_sample.weight", (256, 128, 3, 3)), ("descriptor.up_sample.bias", (128,))]
    e += _bn("descriptor.bn", 128)
    e += _stage("descriptor.layer_out", 256, 128)
    return OrderedDict(e)


# (prefix, cin, cout, stride) of the six two-block stages, in forward order.
STAGES = [
    ("encoder.layer1", 64, 64, 1),
    ("encoder.layer2", 64, 128, 2),
    ("detector.layer", 128, DET_CH, 1),
    ("descriptor.layer_in", 128, 256, 2),
    ("descriptor.layer_out", 256, 128, 1),
]


def conv_macs(h, w, descriptor=True):
    """Algorithmic MACs per frame (convs + the transposed conv only; BN folded,
    element-wise work not counted) -- SURVEY.md section 8(d) / appendix A."""
    def blk(cin, cout, hh, ww, proj):
        m = hh * ww * cout * (cin * 9 + cout)
        if proj:
            m += hh * ww * cout * cin
        return m
    m = (h // 2) * (w // 2) * 64 * 3 * 49
    h4, w4, h8, w8, h16, w16 = h // 4, w // 4, h // 8, w // 8, h // 16, w // 16
    m += blk(64, 64, h4, w4, True) + blk(64, 64, h4, w4, False)
    m += blk(64, 128, h8, w8, True) + blk(128, 128, h8, w8, False)
    m += blk(128, DET_CH, h8, w8, True) + blk(DET_CH, DET_CH, h8, w8, False)
    if descriptor:
        m += blk(128, 256, h16, w16, True) + blk(256, 256, h16, w16, False)
        m += h16 * w16 * 256 * 128 * 9           # ConvTranspose2d: 9 taps per INPUT pixel
        m += blk(256, 128, h8, w8, True) + blk(128, 128, h8, w8, False)
    return m


# ---- the reference's C++ network: superpoint::SPModel (cpp/src/model.cc:4-94, cpp/src/settings.h:19-25) ----
VGG_ENCODER_DIMS = [(1, 64), (64, 64), (64, 128), (128, 128)]   # settings.h:19-22
VGG_DESC_DIM = 256                                               # settings.h:25


def vgg_state_dict_spec():
    """Ordered {name: shape} of SPModel's 24 parameters, as its named_parameters() lists them (the keys of the
    flat dict cpp/src/superpoint.cc:27-55 loads; printed by oracle/_ref/ref_vgg_forward)."""
    e = []
    for i, (cin, cout) in enumerate(VGG_ENCODER_DIMS):
        e += [("encoder_conv%d_a.weight" % i, (cout, cin, 3, 3)), ("encoder_conv%d_a.bias" % i, (cout,)),
              ("encoder_conv%d_b.weight" % i, (cout, cout, 3, 3)), ("encoder_conv%d_b.bias" % i, (cout,))]
    e += [("detector_conv_a.weight", (256, 128, 3, 3)), ("detector_conv_a.bias", (256,)),
          ("detector_conv_b.weight", (65, 256, 1, 1)), ("detector_conv_b.bias", (65,)),
          ("descriptor_conv_a.weight", (256, 128, 3, 3)), ("descriptor_conv_a.bias", (256,)),
          ("descriptor_conv_b.weight", (256, 256, 1, 1)), ("descriptor_conv_b.bias", (256,))]
    return OrderedDict(e)


def vgg_conv_macs(h, w):
    """Algorithmic MACs per frame of SPModel::forward (26.0 G at 640x480)."""
    m, hh, ww = 0, h, w
    for i, (cin, cout) in enumerate(VGG_ENCODER_DIMS):
        m += hh * ww * cout * 9 * (cin + cout)
        if i != 3:
            hh, ww = hh // 2, ww // 2
    m += hh * ww * (256 * 128 * 9 + 65 * 256) + hh * ww * (256 * 128 * 9 + 256 * 256)
    return m
