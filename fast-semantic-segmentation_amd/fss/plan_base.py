"""Tables and small classes shared by the parts of the plan compiler (fss.plan)."""
import torch

from fss import _lib

# storage types of a tensor.  'f32s': split-bfloat16 (FSS_F32_BF16X3) - one 4-byte slot per
# element, every 8 channels stored as their 8 bf16 high parts followed by their 8 bf16 remainders;
# what the float32x3 / mixed modes keep their activations in (the logits stay plain 'f32')
_TORCH_DT = {'f32': torch.float32, 'f32s': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16,
             'u8': torch.uint8}
_CODE = {'f32': _lib.F32, 'f32s': _lib.F32_BF16X3, 'bf16': _lib.BF16, 'f16': _lib.F16}
_EPC = {'f32': 4, 'f32s': 8, 'bf16': 8, 'f16': 8}       # channels per addressable chunk
_ESZ = {'f32': 4, 'f32s': 4, 'bf16': 2, 'f16': 2, 'u8': 1}

# Convolutions (by the scope of their `weights` variable) that the 'mixed' mode computes in half
# precision: the 3x3 convolution of every ResNet bottleneck unit (conv2,
# third_party/dilated_resnet_v1.py:95-97) and the four convolutions of ICNet's full-resolution
# branch (architectures/icnet_architecture.py:213-226).  Chosen on the float32 oracle and on the
# GPU with one group of convolutions at a time in half precision (tools/precision_emulation.py,
# tools/mixed_parity.py, tools/mixed_taps.py): these move the argmax labels least per millisecond
# they save (full-size ICNet-1.0: conv2 of blocks 1+2 / 3 / 4 flip 0.020 / 0.030 / 0.013 % of the
# labels, the branch 0.022 %, all together 0.037 %).  The residual stream of the blocks, the 1x1
# convolutions that read or write it and the whole head stay float with split-bf16 products:
# half precision there costs two to three times as many labels for a fifth of the time saved.
DEFAULT_HALF_SCOPES = (r'/bottleneck_v1/conv2$', r'^Conv(_[123])?$')
# 1x1 convolutions that the 'mixed' mode feeds with half-precision ACTIVATIONS while their weights
# keep float quality (two f16 numbers per weight, two MFMAs per product instead of three): the last
# convolution of a bottleneck unit (conv3, third_party/dilated_resnet_v1.py:98-100), whose input is
# the half-precision convolution before it.  Rounding an activation to 11 bits is an independent
# error per element that averages out over the K axis; rounding a weight is the same error for
# every pixel.  On the float32 oracle, 11-bit activations in every conv3 raise the rms logit error
# of the mode from 1.82e-4 to 1.93e-4 of the logit rms (tools/precision_emulation.py style run),
# where 11-bit weights in the same layers cost several times that.  The two convolutions that read
# the max-pooled output of the first convolution (conv1 and the projection shortcut of the first
# unit) take half-precision activations as well, so that the largest tensor of the backbone and
# its pooled form are written and read at two bytes per element (rms logit error 2.25e-4).  The 1x1
# convolution of a cascade feature fusion (architectures/icnet_architecture.py:243-247) reads its
# branch the same way where nothing else needs that tensor at float quality - the full-resolution
# branch in front of the last fusion (2.29e-4).  In MobileNetV2 the projection of an inverted
# residual (third_party/conv_blocks.py:290-300) reads the depthwise output this way, and the
# expanded tensors on both sides of the depthwise convolution - internal to the unit, six times
# as wide as its input - are half precision (ICNet-MobileNetV2: 3.51e-4 -> 3.71e-4).  The 3x3
# convolution over PSPNet's 4096-channel concatenation (architectures/pspnet_architecture.py:144)
# is two fifths of that model's time at three MFMAs per product: its input is converted to half
# precision in a pass of its own, which a K axis of 36864 repays (PSPNet: 2.25e-4 -> 2.75e-4).
DEFAULT_HALF_ACT_SCOPES = (r'/bottleneck_v1/conv3$', r'/block1/unit_1/bottleneck_v1/(conv1|shortcut)$',
                           r'^CascadeFeatureFusion(_\d+)?/Conv$', r'/expanded_conv(_\d+)?/project$',
                           r'^PSPModule/Conv_\d+$')


class LNode(object):
    """Private, mutable copy of a graph node used during lowering."""

    def __init__(self, kind, inputs, attrs, shape, dtype):
        self.kind, self.inputs, self.attrs = kind, list(inputs), dict(attrs)
        self.shape, self.dtype = tuple(shape), dtype
        self.pos = -1
        self.consumers = []
        self.is_output = False


class Val(object):
    """A materialised tensor: torch buffer + logical shape + physical layout."""

    def __init__(self, buf, shape, dt, phys_c=None, ld=None, coff=0):
        self.buf = buf              # torch tensor [n, H, W, ld] (may be wider / larger than shape)
        self.shape = tuple(shape)   # logical (n, h, w, c)
        self.dt = dt                # 'f32' | 'bf16' | 'u8'
        self.phys_c = phys_c if phys_c is not None else shape[3]
        self.ld = ld if ld is not None else self.phys_c   # pixel stride in elements
        self.coff = coff            # channel offset inside buf

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.coff * self.buf.element_size()

    @property
    def dense(self):
        return self.ld == self.phys_c and self.coff == 0


class Step(object):
    __slots__ = ('kind', 'name', 'fn', 'args', 'flops', 'graph_flops', 'bytes', 'keep', 'stream', 'waits',
                 'signal', 'mpp')

    def __init__(self, kind, name, fn, args, flops=0, nbytes=0, keep=None, graph_flops=None):
        self.kind, self.name, self.fn, self.args = kind, name, fn, args   # args[-1] is the stream
        self.flops, self.bytes, self.keep = flops, nbytes, keep
        # FLOP of the graph nodes this launch stands for, as the graph is written (differs from
        # `flops`, the FLOP executed, where a rewrite lowered the operation count)
        self.graph_flops = flops if graph_flops is None else graph_flops
        self.mpp = 1            # MFMA instructions issued per product (3: split-bf16 arithmetic)
        self.stream = 0         # 0: the plan's stream, 1: the side stream
        self.waits = []         # events its stream waits for before the launch
        self.signal = None      # event recorded on its stream after the launch


# index of the output-pointer arguments of every kernel entry point (everything else
# that points into an activation buffer is read)
_OUT_ARGS = {
    'fss_preprocess_u8': (1,), 'fss_stem_conv_u8': (3,), 'fss_stem2_conv_u8': (5,), 'fss_conv2d_nhwc': (5,), 'fss_conv2d_splitk_nhwc': (5, 7),
    'fss_conv2d_dual_nhwc': (7,), 'fss_conv1x1_chain_nhwc': (11, 17), 'fss_conv3x3_c64_nhwc': (4,), 'fss_conv3x3_c128_nhwc': (4,),
    'fss_conv2d_dual_head_nhwc': (13,), 'fss_zoom2_border_gather': (1, 2, 3), 'fss_zoom2_conv_commute': (4,),
    'fss_dwconv2d_nhwc': (3,), 'fss_subsample_nhwc': (1,), 'fss_maxpool_nhwc': (1,),
    'fss_avgpool_bins_nhwc': (1,), 'fss_avgpool_pyramid_nhwc': (1, 7), 'fss_resize_bilinear_nhwc': (1,), 'fss_pyramid_add_nhwc': (5,),
    'fss_add_act': (2,), 'fss_to_half': (1,), 'fss_copy_channels': (2,), 'fss_upsample_argmax': (1, 2),
    'fss_label_lut': (1,),
}
