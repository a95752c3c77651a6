"""oracle/models.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional CPU restatement of the three backbones' forward passes with fast weights, written
straight from the reference's forward() bodies.  `base` is the module's own {name: tensor}
(parameters + buffers); `fast` is the inner-loop dict or None.  Where the reference hands the fast
dict to a sub-module, W() reads it; where it does not, the module's own tensors are used.

Pinned by tests/golden/*: fixtures produced by importing the reference's MetaNetwork / MetaCAIN /
MetaVoxelFlow here (oracle/gen_golden.py) on the seeded weights of meta-interpolation_amd/synthetic.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import torch
import torch.nn.functional as F

from . import torch_ops as O


def _W(base, fast):
    def get(name):
        if fast is not None:
            return fast[name]           # KeyError here == the reference's KeyError on params[...]
        return base[name]
    return get


def _up2(x):  # torch.nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)


# ---------------------------------------------------------------------------------------------
# SepConv                                                   sepconv/model.py:252-349, :172-194
# ---------------------------------------------------------------------------------------------
def sepconv_forward(f0, f1, base, fast=None, sepconv_op=None):
    op = sepconv_op or O.SepconvCPU.apply
    own = _W(base, None)
    fw = _W(base, fast)           # moduleConv*/moduleDeconv* get the fast dict (:276-306)

    def basic(x, prefix, get):    # Basic(): 3 x (conv3x3 + ReLU) at indices 0, 2, 4 (:172-180)
        for i in (0, 2, 4):
            x = F.relu(F.conv2d(x, get('%s.%d.weight' % (prefix, i)), get('%s.%d.bias' % (prefix, i)), 1, 1))
        return x

    def upsample(x, prefix):      # moduleUpsampleN: own parameters always (:292,297,302,307)
        return F.relu(F.conv2d(_up2(x), own(prefix + '.1.weight'), own(prefix + '.1.bias'), 1, 1))

    def subnet(x, prefix):        # Subnet(): own parameters always (:346-347 pass no params)
        for i in (0, 2, 4):
            x = F.relu(F.conv2d(x, own('%s.%d.weight' % (prefix, i)), own('%s.%d.bias' % (prefix, i)), 1, 1))
        return F.conv2d(_up2(x), own(prefix + '.7.weight'), own(prefix + '.7.bias'), 1, 1)

    width, height = f0.size(3), f0.size(2)
    pw, ph = 25 + width + 25, 25 + height + 25                     # :255-260
    if pw != ((pw >> 7) << 7):
        pw = (((pw >> 7) + 1) << 7)
    if ph != ((ph >> 7) << 7):
        ph = (((ph >> 7) + 1) << 7)
    pad_in = [25, pw - 25 - width, 25, ph - 25 - height]           # :261-263
    p0 = F.pad(f0, pad_in, mode='replicate')
    p1 = F.pad(f1, pad_in, mode='replicate')
    join = torch.cat([p0, p1], 1)

    c1 = basic(join, 'moduleConv1', fw)
    c2 = basic(F.avg_pool2d(c1, 2, 2), 'moduleConv2', fw)
    c3 = basic(F.avg_pool2d(c2, 2, 2), 'moduleConv3', fw)
    c4 = basic(F.avg_pool2d(c3, 2, 2), 'moduleConv4', fw)
    c5 = basic(F.avg_pool2d(c4, 2, 2), 'moduleConv5', fw)
    x = upsample(basic(F.avg_pool2d(c5, 2, 2), 'moduleDeconv5', fw), 'moduleUpsample5') + c5
    x = upsample(basic(x, 'moduleDeconv4', fw), 'moduleUpsample4') + c4
    x = upsample(basic(x, 'moduleDeconv3', fw), 'moduleUpsample3') + c3
    x = upsample(basic(x, 'moduleDeconv2', fw), 'moduleUpsample2') + c2

    rp = lambda t: F.pad(t, [25, 25, 25, 25], mode='replicate')   # modulePad (:244-245)
    d1 = op(rp(p0).contiguous(), subnet(x, 'moduleVertical1'), subnet(x, 'moduleHorizontal1'))
    d2 = op(rp(p1).contiguous(), subnet(x, 'moduleVertical2'), subnet(x, 'moduleHorizontal2'))
    out = d1 + d2
    return out[:, :, 25:25 + height, 25:25 + width]                # modulePaddingOutput (:264-266, :349)


# ---------------------------------------------------------------------------------------------
# CAIN                     cain/model.py:70-94; model_utils.py:11-28, :821-848, :931-1053
# ---------------------------------------------------------------------------------------------
def cain_forward(x1, x2, base, fast=None, pixel_shuffle=None):
    ps = pixel_shuffle or O.pixel_shuffle
    get = _W(base, fast)          # every tensor of CAIN is routed through the fast dict

    def sub_mean(x):
        m = x.mean(2, keepdim=True).mean(3, keepdim=True)
        return x - m, m

    def convnorm(x, prefix):      # MetaConvNorm: ReflectionPad2d(1) + conv3x3 padding 0
        return F.conv2d(F.pad(x, [1, 1, 1, 1], mode='reflect'), get(prefix + '.conv.weight'),
                        get(prefix + '.conv.bias'), 1, 0)

    def rcab(x, prefix):          # MetaRCAB body: ConvNorm, LeakyReLU(0.2), ConvNorm, CALayer; + x
        y = convnorm(x, prefix + '.body.0')
        y = F.leaky_relu(y, 0.2)
        y = convnorm(y, prefix + '.body.2')
        a = F.adaptive_avg_pool2d(y, 1)
        a = F.relu(F.conv2d(a, get(prefix + '.body.3.conv_du.0.weight'), get(prefix + '.body.3.conv_du.0.bias')))
        a = torch.sigmoid(F.conv2d(a, get(prefix + '.body.3.conv_du.2.weight'), get(prefix + '.body.3.conv_du.2.bias')))
        return y * a + x

    x1, m1 = sub_mean(x1)
    x2, m2 = sub_mean(x2)
    w, h = x1.size(3), x1.size(2)                                   # InOutPaddings (:17-28)
    pw = 0 if w == ((w >> 7) << 7) else (((w >> 7) + 1) << 7) - w
    ph = 0 if h == ((h >> 7) << 7) else (((h >> 7) + 1) << 7) - h
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    x1 = F.pad(x1, pad, mode='reflect')
    x2 = F.pad(x2, pad, mode='reflect')

    root = 'encoder.interpolate'
    x = torch.cat([ps(x1, 1 / 8), ps(x2, 1 / 8)], dim=1)
    x = F.conv2d(x, get(root + '.headConv.weight'), get(root + '.headConv.bias'), 1, 1)
    res = x
    for g in range(5):
        gin = res
        for b in range(12):
            res = rcab(res, '%s.body.%d.body.%d' % (root, g, b))
        res = convnorm(res, '%s.body.%d.body.12' % (root, g)) + gin
    res = res + x
    out = F.conv2d(res, get(root + '.tailConv.weight'), get(root + '.tailConv.bias'), 1, 1)
    out = ps(out, 8)
    out = out[:, :, pad[2]:pad[2] + h, pad[0]:pad[0] + w]
    return out + (m1 + m2) / 2


# ---------------------------------------------------------------------------------------------
# VoxelFlow                               voxelflow/core/models/voxel_flow.py:357-509
# ---------------------------------------------------------------------------------------------
def voxelflow_forward(x0, x1, base, fast=None, warp=None):
    warp = warp or O.voxel_warp_blend
    get = _W(base, fast)          # convs read the fast dict; BN layers use their own tensors (:379,...)

    def block(x, name, k):
        x = F.conv2d(x, get(name + '.weight'), None, 1, k // 2)
        x = F.batch_norm(x, base[name + '_bn.running_mean'], base[name + '_bn.running_var'],
                         base[name + '_bn.weight'], base[name + '_bn.bias'], False, 0.9997, 1e-5)
        return F.relu(x)

    x = torch.cat([x0, x1], dim=1)
    w, h = x.size(3), x.size(2)
    pw = 0 if w == ((w >> 6) << 6) else (((w >> 6) + 1) << 6) - w
    ph = 0 if h == ((h >> 6) << 6) else (((h >> 6) + 1) << 6) - h
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    inp = F.pad(x, pad, mode='reflect') if (pw or ph) else x
    up = lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)

    c1 = block(inp, 'conv1', 5)
    c2 = block(F.max_pool2d(c1, 2, 2), 'conv2', 5)
    c3 = block(F.max_pool2d(c2, 2, 2), 'conv3', 3)
    y = block(F.max_pool2d(c3, 2, 2), 'bottleneck', 3)
    y = block(torch.cat([up(y), c3], 1), 'deconv1', 3)
    y = block(torch.cat([up(y), c2], 1), 'deconv2', 5)
    y = block(torch.cat([up(y), c1], 1), 'deconv3', 5)
    x3 = torch.tanh(F.conv2d(y, get('conv4.weight'), get('conv4.bias'), 1, 2))
    out = warp(inp, x3)
    return out[:, :, pad[2]:pad[2] + h, pad[0]:pad[0] + w]


# ---------------------------------------------------------------------------------------------
# RRIN                                                    rrin/model.py:69-131, rrin/unet.py:101-208
# ---------------------------------------------------------------------------------------------
def _lrelu(x):
    return F.leaky_relu(x, negative_slope=0.1)


def _reflect_to(x, shift):
    """Reflection pad to a multiple of 2**shift; returns (padded, crop function)."""
    w, h = x.size(3), x.size(2)
    pw = 0 if w == ((w >> shift) << shift) else (((w >> shift) + 1) << shift) - w
    ph = 0 if h == ((h >> shift) << shift) else (((h >> shift) + 1) << shift) - h
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    crop = lambda t: t[:, :, pad[2]:pad[2] + h, pad[0]:pad[0] + w]
    return (lambda t: F.pad(t, pad, mode='reflect') if (pw or ph) else t), crop


def _rrin_unet(x, get, name, depth):
    """MetaUNet.forward (unet.py:130-158): conv blocks + avg-pool down, midconv, (x2 bilinear, conv, cat, conv block) up."""
    conv = lambda t, n: F.conv2d(t, get(n + '.weight'), get(n + '.bias'), 1, 1)
    block = lambda t, n: _lrelu(conv(_lrelu(conv(t, n + '.block.0')), n + '.block.2'))
    bridges = []
    for i in range(depth):
        x = block(x, '%s.down_path.%d' % (name, i))
        if i != depth - 1:
            bridges.append(x)
            x = F.avg_pool2d(x, 2)
    x = _lrelu(conv(x, name + '.midconv'))
    for i in range(depth - 1):
        up = conv(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), '%s.up_path.%d.up.1' % (name, i))
        bridge = bridges[-i - 1]
        dy, dx = (bridge.size(2) - up.size(2)) // 2, (bridge.size(3) - up.size(3)) // 2
        x = block(torch.cat((up, bridge[:, :, dy:dy + up.size(2), dx:dx + up.size(3)]), 1), '%s.up_path.%d.conv_block' % (name, i))
    return conv(x, name + '.last')


def rrin_forward(x0, x1, base, fast=None, warp=None, t=0.5):
    warp = warp or O.flow_warp
    get = _W(base, fast)
    pad, crop = _reflect_to(x0, 7)
    x0, x1 = pad(x0), pad(x1)
    x = torch.cat((x0, x1), 1)
    flow = _rrin_unet(x, get, 'Flow_L', 5)
    f01, f10 = flow[:, :2], flow[:, 2:4]
    ft0 = -(1 - t) * t * f01 + t * t * f10
    ft1 = (1 - t) * (1 - t) * f01 - t * (1 - t) * f10
    res = _rrin_unet(torch.cat((ft0, ft1, x), 1), get, 'refine_flow', 4)
    ft0, ft1 = ft0 + res[:, :2], ft1 + res[:, 2:4]
    xt0, xt1 = warp(x0, ft0), warp(x1, ft1)
    mask = torch.sigmoid(_rrin_unet(torch.cat((ft0, ft1, x, xt0, xt1), 1), _W(base, None), 'Mask', 4))   # own weights: :104
    w0, w1 = (1 - t) * mask[:, 0:1], t * mask[:, 1:2]
    blend = (w0 * xt0 + w1 * xt1) / (w0 + w1 + 1e-8)
    final = _rrin_unet(torch.cat((x0, x1, blend), 1), get, 'final', 4) + blend
    return crop(final.clamp(0, 1))


# ---------------------------------------------------------------------------------------------
# Super SloMo                                                     superslomo/model.py:457-646
# ---------------------------------------------------------------------------------------------
def _slomo_unet(x, get, name):
    """MetaUNet.forward (:499-545): 7x7, 7x7, five (avg-pool, k x k, k x k) downs, five (x2, 3x3, cat skip, 3x3) ups, 3x3."""
    conv = lambda t, n, k: _lrelu(F.conv2d(t, get('%s.%s.weight' % (name, n)), get('%s.%s.bias' % (name, n)), 1, (k - 1) // 2))
    x = conv(x, 'conv1', 7)
    skips = [conv(x, 'conv2', 7)]
    for i, k in enumerate((5, 3, 3, 3, 3)):
        y = F.avg_pool2d(skips[-1] if i < 5 else x, 2)
        y = conv(conv(y, 'down%d.conv1' % (i + 1), k), 'down%d.conv2' % (i + 1), k)
        skips.append(y)
    x = skips.pop()
    for i in range(5):
        x = conv(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), 'up%d.conv1' % (i + 1), 3)
        x = conv(torch.cat((x, skips.pop()), 1), 'up%d.conv2' % (i + 1), 3)
    return conv(x, 'conv3', 3)


def superslomo_forward(I0, I1, base, fast=None, warp=None, ind=3):
    """Returns the interpolated (mean-subtracted) frame; the extra outputs of the reference only feed its 'Super' loss."""
    import numpy as np
    warp = warp or O.flow_warp
    get = _W(base, fast)
    t = float(np.linspace(0.125, 0.875, 7)[ind])                         # :308, getFlowCoeff / getWarpCoeff :310-382
    c00 = c11 = float(np.float32(-(1 - t) * t))
    c01, c10 = float(np.float32(t * t)), float(np.float32((1 - t) * (1 - t)))
    pad, crop = _reflect_to(I0, 6)
    I0, I1 = pad(I0), pad(I1)
    flows = _slomo_unet(torch.cat((I0, I1), 1), get, 'flowComp')
    F01, F10 = flows[:, :2], flows[:, 2:]
    Ft0 = c00 * F01 + c01 * F10
    Ft1 = c10 * F01 + c11 * F10
    g0, g1 = warp(I0, Ft0), warp(I1, Ft1)
    intrp = _slomo_unet(torch.cat((I0, I1, F01, F10, Ft1, Ft0, g1, g0), 1), get, 'arbTimeFlowIntrp')
    Ft0f, Ft1f = intrp[:, :2] + Ft0, intrp[:, 2:4] + Ft1
    V0 = torch.sigmoid(intrp[:, 4:5])
    V1 = 1 - V0
    g0f, g1f = warp(I0, Ft0f), warp(I1, Ft1f)
    C0, C1 = float(np.float32(1 - t)), float(np.float32(t))
    return crop((C0 * V0 * g0f + C1 * V1 * g1f) / (C0 * V0 + C1 * V1))


FORWARD = {'sepconv': sepconv_forward, 'cain': cain_forward, 'voxelflow': voxelflow_forward, 'rrin': rrin_forward,
           'superslomo': superslomo_forward}
