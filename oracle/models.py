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


FORWARD = {'sepconv': sepconv_forward, 'cain': cain_forward, 'voxelflow': voxelflow_forward}
