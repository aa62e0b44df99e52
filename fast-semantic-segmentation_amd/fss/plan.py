"""Lowers an `fss.graph` graph to launches of the gfx950 kernels and runs it.

compile = (1) copy the live sub-graph into a private node list and rewrite it,
(2) lower node by node with fusion, assigning device buffers, (3) fold batch
norm into / pack the weights, (4) one eager run, (5) capture the launch
sequence into a hipGraph that `launch()` replays.

Fusions (each replaces a chain of reference ops by one kernel):
  * placeholder -> pad_to_bounding_box -> to_float -> channel affine   => fss_preprocess_u8
  * ... that chain [-> exact 1/f align_corners resize] -> first conv2d + batch norm
    + activation                                                       => fss_stem_conv_u8
  * conv2d + folded batch norm + activation [+ add(other) [+ relu]]    => fss_conv2d_nhwc
  * x + resize(avg_pool(x)) + ... of the fast pyramid pooling module   => fss_pyramid_add_nhwc
  * resize(logits) -> argmax                                           => fss_upsample_argmax
  * lut -> crop                                                        => fss_label_lut
Rewrites:
  * a same-size resize is dropped;
  * the 1x1 biased classifier is moved in front of the bilinear zoom that
    feeds it (exact up to rounding: bilinear weights sum to one), so the zoom
    runs on num_classes channels instead of hundreds.
Device memory, streams and host<->device copies go through torch; all
arithmetic goes through libfss_gfx950.so.
"""
import bisect
import ctypes
import json
import os
import time

import numpy as np
import torch

from fss import _lib
from fss._lib import lib, check

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
# branch in front of the last fusion (2.29e-4).
DEFAULT_HALF_ACT_SCOPES = (r'/bottleneck_v1/conv3$', r'/block1/unit_1/bottleneck_v1/(conv1|shortcut)$',
                           r'^CascadeFeatureFusion(_\d+)?/Conv$')


def f32_to_bf16_bits(a):
    """float32 ndarray -> uint16 ndarray of bfloat16 bits (round to nearest even)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


def f32_to_f16_bits(a):
    """float32 ndarray -> uint16 ndarray of IEEE half bits (round to nearest even, saturating)."""
    a = np.clip(np.ascontiguousarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).view(np.uint16)


def f32_split_f16_rows(Wp):
    """Packed float32 weights [cout_pad][k_pad] -> uint16 [cout_pad][2 * k_pad]: every row's
    IEEE-half high parts followed by its remainders (the FSS_F16_W2 convolutions)."""
    hi = f32_to_f16_bits(Wp)
    lo = f32_to_f16_bits(np.asarray(Wp, np.float32) - hi.view(np.float16).astype(np.float32))
    return np.ascontiguousarray(np.concatenate([hi, lo], axis=1))


def stem_pack_f16x2(W, mean, scale, cout_pad, k_pad):
    """Weights of fss_stem_conv_u8 in its FSS_F32_BF16X3 mode from the filter W [kh,kw,3,cout]
    (batch norm folded in): the kernel multiplies the raw frame bytes and a 1 per pixel inside the
    padded image, so scale goes into the three channel weights and -sum_c W_c mean_c scale_c into
    the fourth; returned as uint16 [2 * cout_pad][k_pad]: IEEE-half high parts, then remainders."""
    kh, kw, _, cout = W.shape
    W = np.asarray(W, np.float64)
    ms = np.asarray(mean, np.float64) * np.asarray(scale, np.float64)
    W4 = np.zeros((kh, kw, 4, cout))
    W4[:, :, :3] = W * np.asarray(scale, np.float64)[None, None, :, None]
    W4[:, :, 3] = -(W * ms[None, None, :, None]).sum(axis=2)
    Wp = np.zeros((cout_pad, k_pad))
    Wp[:cout, :kh * kw * 4] = W4.transpose(3, 0, 1, 2).reshape(cout, -1)
    hi = f32_to_f16_bits(Wp.astype(np.float32))
    lo = f32_to_f16_bits((Wp - hi.view(np.float16).astype(np.float64)).astype(np.float32))
    return np.concatenate([hi, lo])


def split_bf16_decode(raw):
    """float32 ndarray holding split-bfloat16 groups along its last axis (a multiple of 8 slots:
    8 bf16 high parts, then 8 bf16 remainders, per 8 elements) -> the float32 values."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    u = raw.view(np.uint16).reshape(raw.shape[:-1] + (raw.shape[-1] // 8, 2, 8)).astype(np.uint32) << 16
    f = u.view(np.float32)
    return (f[..., 0, :] + f[..., 1, :]).reshape(raw.shape)


def split_bf16_encode(a):
    """float32 ndarray (last axis a multiple of 8) -> float32-typed array holding its
    split-bfloat16 form (same shape and bytes per element)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return f32_split_bf16_rows(a.reshape(-1, a.shape[-1])).view(np.float32).reshape(a.shape)


def f32_split_bf16_rows(wp):
    """Packed float32 weights [cout_pad, k_pad] -> the FSS_F32_BF16X3 form: per 8 consecutive k
    values of a row their 8 bfloat16 high parts, then their 8 bfloat16 remainders (uint16
    array of shape [cout_pad, 2 * k_pad], the same bytes as the float32 matrix)."""
    wp = np.ascontiguousarray(wp, dtype=np.float32)
    hi = f32_to_bf16_bits(wp)
    hi_f = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = f32_to_bf16_bits(wp - hi_f)
    r, k = wp.shape
    out = np.empty((r, k // 8, 2, 8), np.uint16)
    out[:, :, 0, :] = hi.reshape(r, k // 8, 8)
    out[:, :, 1, :] = lo.reshape(r, k // 8, 8)
    return out.reshape(r, 2 * k)


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
    'fss_conv2d_dual_nhwc': (7,), 'fss_conv1x1_chain_nhwc': (11, 17), 'fss_conv3x3_c64_nhwc': (4,),
    'fss_conv2d_dual_head_nhwc': (13,), 'fss_zoom2_border_gather': (1, 2, 3), 'fss_zoom2_conv_commute': (4,),
    'fss_dwconv2d_nhwc': (3,), 'fss_subsample_nhwc': (1,), 'fss_maxpool_nhwc': (1,),
    'fss_avgpool_bins_nhwc': (1,), 'fss_avgpool_pyramid_nhwc': (1, 7), 'fss_resize_bilinear_nhwc': (1,), 'fss_pyramid_add_nhwc': (5,),
    'fss_add_act': (2,), 'fss_copy_channels': (2,), 'fss_upsample_argmax': (1, 2),
    'fss_label_lut': (1,),
}


class CompiledGraph(object):
    """Executable form of a graph for fixed shapes, weights and compute dtype.

    outputs: graph tensors to make fetchable.  dtype: 'float32' | 'bfloat16'
    (storage + MFMA input type of activations; logits and everything after the
    classifier stay float32)."""

    def __init__(self, graph, outputs, weights, dtype='bfloat16', device=None,
                 commute_classifier=True, use_hip_graph=True, autotune=True, mfma_edge=None,
                 fuse_stem=True, streams=2, fuse_pyramid_pool=True, fuse_stem2=True, fuse_dual=True, fuse_chain=True, commute_pyramid=None, resident_3x3=True,
                 fuse_head=True, collapse_zooms=True, commute_zoom=None, fuse_zoom_residual=None,
                 ksplit=False, half_scopes=None, half_act_scopes=None, reuse_buffers=None):
        if not torch.cuda.is_available():
            raise RuntimeError('fss.plan needs a ROCm GPU; there is no CPU execution path')
        # 'float32x3': float32 tensors everywhere; the convolutions form their products on the
        # bf16 matrix pipe from bf16 hi/lo splits of both operands (FSS_F32_BF16X3)
        # 'mixed': float32x3, except that the convolutions named by `half_scopes` (regular
        # expressions searched in the scope of a convolution's weights; DEFAULT_HALF_SCOPES) run
        # on half-precision tensors with one f16 MFMA per product.  A tensor is stored in half
        # precision when every consumer is such a convolution; a convolution computes in the type
        # its input is stored in.
        self.mixed = dtype == 'mixed'
        self.split = dtype in ('float32x3', 'f32x3', 'mixed')
        self.dt = {'float32': 'f32', 'bfloat16': 'bf16', 'f32': 'f32', 'bf16': 'bf16',
                   'float32x3': 'f32s', 'f32x3': 'f32s', 'mixed': 'f32s'}[dtype]
        import re
        self.half_re = [re.compile(p) for p in (DEFAULT_HALF_SCOPES if half_scopes is None else half_scopes)]
        # `half_act_scopes` (DEFAULT_HALF_ACT_SCOPES): 1x1 convolutions that read half-precision
        # activations with float-quality weights, so that the tensor in front of them is stored in
        # half precision too
        if half_act_scopes is None and 'FSS_HALF_ACT_SCOPES' in os.environ:      # experiments: '' = none
            half_act_scopes = [p for p in os.environ['FSS_HALF_ACT_SCOPES'].split(',') if p]
        self.half_act_re = [re.compile(p) for p in
                            (DEFAULT_HALF_ACT_SCOPES if half_act_scopes is None else half_act_scopes)]
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.weights = weights
        self.outputs = list(outputs)
        self.stream = torch.cuda.Stream(device=self.device)
        self.sh = ctypes.c_void_p(self.stream.cuda_stream)
        self.steps = []
        self.step_fname = []
        self.vals = {}          # LNode -> Val
        self.lazy = {}          # LNode -> (tag, info..., materialise closure)
        self.keepalive = []     # device-side constants (packed weights, biases, tables)
        self.temps = []         # activation buffers between launches that stand for one graph node
        self.workspaces = []    # scratch tensors a launch both writes and reads
        self.input_val = None
        self.graph_exec = None
        self.param_bytes = 0
        # MFMA shape class of the conv kernel: 16 (v_mfma_*_16x16*) or 32 (v_mfma_*_32x32*)
        self.mfma_edge = int(mfma_edge or os.environ.get('FSS_MFMA_EDGE',
                                                         16 if (self.dt == 'bf16' or self.split) else 32))
        if self.mfma_edge not in (16, 32):
            raise ValueError('mfma_edge must be 16 or 32')
        self.fuse_stem = fuse_stem
        # split-K launches for convolutions with few output tiles and a long K axis (one or two
        # frames per launch): lower latency, but the float additions run in another order, so the
        # bits then depend on the batch size; off unless asked for
        self.ksplit = ksplit
        self.fuse_pyramid_pool = fuse_pyramid_pool
        self.fuse_stem2 = fuse_stem2
        self.fuse_dual = fuse_dual
        self.fuse_chain = fuse_chain and os.environ.get('FSS_FUSE_CHAIN', '1') != '0'
        self.commute_pyramid = commute_pyramid
        self.resident_3x3 = resident_3x3 and os.environ.get('FSS_RESIDENT_3X3', '1') != '0'
        self.fuse_head = fuse_head
        self.collapse_zooms = collapse_zooms
        self.commute_zoom = commute_zoom
        self.fuse_zoom_residual = (os.environ.get('FSS_FUSE_ZOOM_RESIDUAL', '1') != '0'
                                   if fuse_zoom_residual is None else fuse_zoom_residual)
        self.dual = {}          # 1x1 conv node -> the conv whose output it is added to (one launch)
        self.two_streams = False
        self.release_after = None
        self.stream_choice = 'single: not requested'
        self.events = []
        self.tuned_now = 0      # layer shapes timed by this plan (0: all came from the table)
        # Activation memory: tensors whose lifetimes (first to last launch that touches them) do
        # not overlap share device memory.  Off: every tensor has a buffer of its own.
        self.reuse_buffers = (os.environ.get('FSS_REUSE_BUFFERS', '1') != '0'
                              if reuse_buffers is None else reuse_buffers)
        self._allocs = []       # tensors handed out by _alloc, in call order
        self._mem_plan = None   # per _alloc call: (bytes, first step, last step, stream, shareable)
        self._blocks = []       # device blocks behind the shared tensors: [tensor, tenant's last step, stream]
        self.activation_bytes = self.activation_bytes_unshared = 0
        with torch.cuda.device(self.device):
            self._import(graph, outputs, commute_classifier)
            if self.reuse_buffers:
                self._plan_memory(streams > 1 and use_hip_graph)
            self._lower()
            self._dependencies()                  # every pointer of every launch must resolve
            torch.cuda.synchronize(self.device)   # constants were uploaded on torch's stream
            if autotune:
                self._autotune_convs()
            self.run_eager()
            self.sync()
            if use_hip_graph:
                if streams > 1:
                    self._setup_streams()
                self._capture()

    # ---------------------------------------------------- import + rewrites
    def _import(self, graph, outputs, commute_classifier):
        live, stack = set(), [t.node.index for t in outputs]
        while stack:
            i = stack.pop()
            if i not in live:
                live.add(i)
                stack.extend(t.node.index for t in graph.nodes[i].inputs)
        m = {}
        order = []
        for i in sorted(live):
            nd = graph.nodes[i]
            ln = LNode(nd.kind, [m[t.node.index] for t in nd.inputs], nd.attrs,
                       nd.output.shape, nd.output.dtype)
            m[i] = ln
            order.append(ln)
        self.out_nodes = {t.node.index: m[t.node.index] for t in outputs}

        def replace_uses(old, new):
            for ln in order:
                ln.inputs = [new if x is old else x for x in ln.inputs]
            for k, v in self.out_nodes.items():
                if v is old:
                    self.out_nodes[k] = new

        def n_uses(x):
            return sum(1 for ln in order for y in ln.inputs if y is x) + \
                sum(1 for v in self.out_nodes.values() if v is x)

        # same-size resize -> identity
        for ln in list(order):
            if ln.kind == 'resize' and ln.shape[1:3] == ln.inputs[0].shape[1:3]:
                replace_uses(ln, ln.inputs[0])
                order.remove(ln)
        # classifier in front of the zoom that feeds it
        if commute_classifier:
            for ln in list(order):
                a = ln.attrs
                if not (ln.kind == 'conv2d' and a['bn'] is None and a['activation'] is None
                        and (a['kh'], a['kw'], a['stride'], a['rate']) == (1, 1, 1, 1)):
                    continue
                r = ln.inputs[0]
                if not (r.kind == 'resize' and r.attrs['align_corners'] and n_uses(r) == 1):
                    continue
                x = r.inputs[0]
                conv = LNode('conv2d', [x], a, x.shape[:3] + (ln.shape[3],), ln.dtype)
                zoom = LNode('resize', [conv], r.attrs, ln.shape, ln.dtype)
                at = order.index(ln)
                order[at:at + 1] = [conv, zoom]
                order.remove(r)
                replace_uses(ln, zoom)
        self.order = order
        for p, ln in enumerate(order):
            ln.pos = p
            ln.consumers = []
        for ln in order:
            for x in ln.inputs:
                x.consumers.append(ln)
        for v in self.out_nodes.values():
            v.is_output = True

    # ------------------------------------------------------------------ utils
    def _alloc(self, shape, dt, phys_c=None):
        """Device tensor [n,h,w,phys_c] of an activation.  With a memory plan (second lowering
        pass) a tensor whose lifetime is known may take over the block of one that is dead by
        the time of its first launch; the block then counts as ONE buffer in the dependency
        analysis, so write-after-read and write-after-write between its tenants are ordered like
        any other hazard."""
        n, h, w, c = shape
        full = (n, h, w, phys_c if phys_c is not None else c)
        tdt = _TORCH_DT[dt]
        k = len(self._allocs)
        if self._mem_plan is None:
            t = torch.empty(full, dtype=tdt, device=self.device)
            self._allocs.append(t)
            return t
        nbytes, first, last, stream, shareable = self._mem_plan[k]
        need = int(np.prod(full)) * torch.empty((), dtype=tdt).element_size()
        if need != nbytes:
            raise RuntimeError('the second lowering pass asks for other buffers than the first')
        self.activation_bytes_unshared += need
        blk = None
        if shareable:
            # the smallest free block that holds it and is at most twice its size
            best = None
            for b in self._blocks:
                if b[2] == stream and b[1] < first and need <= b[0].numel() <= 2 * need:
                    if best is None or b[0].numel() < best[0].numel():
                        best = b
            if best is not None:
                best[1] = last
                blk = best[0]
        if blk is None:
            blk = torch.empty((max(need, 1) + 255) // 256 * 256, dtype=torch.uint8, device=self.device)
            self.activation_bytes += blk.numel()
            self._blocks.append([blk, last if shareable else 1 << 60, stream])
        t = blk[:need].view(tdt).view(full)
        self._allocs.append(t)
        return t

    def _reset_lowering(self):
        self.steps, self.step_fname = [], []
        self.vals, self.lazy, self.dual = {}, {}, {}
        self.keepalive, self.temps, self.workspaces = [], [], []
        self.input_val = None
        self.param_bytes = 0
        self._allocs = []

    def _plan_memory(self, two_streams):
        """First lowering pass: every activation in a buffer of its own, nothing launched.  The
        launches it emits say which of them touch which buffer - the ground truth whatever was
        fused, deferred or aliased on the way - hence every buffer's first and last launch and
        the stream(s) it is used on.  A buffer may later share memory only if it is no output,
        is touched by at least one launch, and only by launches of one stream."""
        self._lower()
        allocs = self._allocs
        starts = sorted((t.data_ptr(), k) for k, t in enumerate(allocs))
        first = [None] * len(allocs)
        last = [None] * len(allocs)
        streams = [set() for _ in allocs]
        if two_streams and self._assign_streams(None):
            step_stream = [st.stream for st in self.steps]
        else:
            step_stream = [0] * len(self.steps)
        keys = [a for a, _ in starts]
        for i, (st, fname) in enumerate(zip(self.steps, self.step_fname)):
            argtypes = _lib.PROTOTYPES[fname][1]
            for ai, arg in enumerate(st.args[:-1]):
                if isinstance(arg, ctypes.Array) and getattr(arg, '_type_', None) is ctypes.c_void_p:
                    ptrs = [q for q in arg if q]
                elif argtypes[ai] is ctypes.c_void_p and isinstance(arg, int) and arg:
                    ptrs = [arg]
                else:
                    continue
                for q in ptrs:
                    j = bisect.bisect_right(keys, q) - 1
                    if j < 0:
                        continue
                    k = starts[j][1]
                    t = allocs[k]
                    if q < t.data_ptr() + t.numel() * t.element_size():
                        first[k] = i if first[k] is None else first[k]
                        last[k] = i
                        streams[k].add(step_stream[i])
        pinned = {self.vals[ln].buf.data_ptr() for ln in self.out_nodes.values() if ln in self.vals}
        plan = []
        for k, t in enumerate(allocs):
            ok = first[k] is not None and len(streams[k]) == 1 and t.data_ptr() not in pinned
            plan.append((t.numel() * t.element_size(), first[k] if ok else 0, last[k] if ok else 1 << 60,
                         min(streams[k]) if streams[k] else 0, ok))
        for e in self.events:
            lib.fss_event_destroy(e)
        self.events = []
        self._reset_lowering()
        del allocs, starts
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()
        self._mem_plan = plan

    def _in_code(self, dt):
        """in_dtype code of a convolution whose input is stored as `dt`."""
        return _CODE[dt]

    def _pack(self, Wp, dt):
        """Packed float32 weight matrix -> device tensor in the form a convolution reading `dt`
        tensors takes (bf16 / f16 bits; float, or its split-bf16 rows in the float32x3 modes)."""
        if dt == 'bf16':
            dev = torch.from_numpy(f32_to_bf16_bits(Wp).view(np.int16)).to(self.device)
        elif dt == 'f16':
            dev = torch.from_numpy(f32_to_f16_bits(Wp).view(np.int16)).to(self.device)
        elif dt == 'f16w':
            dev = torch.from_numpy(f32_split_f16_rows(Wp).view(np.int16)).to(self.device)
        elif dt == 'f32s':
            dev = torch.from_numpy(f32_split_bf16_rows(Wp).view(np.int16)).to(self.device)
        else:
            dev = torch.from_numpy(np.ascontiguousarray(Wp, np.float32)).to(self.device)
        self.keepalive.append(dev)
        return dev

    def _is_half_conv(self, ln):
        if not self.mixed or ln.kind != 'conv2d':
            return False
        scope = ln.attrs['weights'].rsplit('/', 1)[0]
        return any(r.search(scope) for r in self.half_re)

    def _takes_half_acts(self, ln):
        """Mixed mode: `ln` is a 1x1 convolution that reads half-precision activations against
        weights kept as two f16 numbers each (emitted as a two-input launch whose second input is
        the first one again and whose second filter holds the weights' remainders)."""
        if not self.mixed or ln.kind != 'conv2d' or self._is_half_conv(ln):
            return False
        a = ln.attrs
        scope = a['weights'].rsplit('/', 1)[0]
        if not any(r.search(scope) for r in self.half_act_re):
            return False
        if ((a['kh'], a['kw'], a['stride'], a['rate'], a['pad_top'], a['pad_left']) != (1, 1, 1, 1, 0, 0)
                or a['biases'] is not None or ln.inputs[0].shape[3] % 64 or ln.shape[:3] != ln.inputs[0].shape[:3]):
            return False
        if not self.fuse_dual:
            return True
        # a convolution that shares a launch with another one (projection shortcut + last 1x1):
        # only when both are named, so that both inputs are half precision
        mate = self._dual_partner(ln)
        add = self._only_consumer(ln, 'add')
        if mate is None and add is not None:
            for other in add.inputs:
                if other is not ln and other.kind == 'conv2d' and self._dual_partner(other) is ln:
                    mate = other
        if mate is None:
            return ln not in self.dual
        if mate.pos < ln.pos and self._zoom_source(mate) is not None:
            return True     # the mate runs in front of its zoom, in launches of its own
        ms = mate.attrs['weights'].rsplit('/', 1)[0]
        return (any(r.search(ms) for r in self.half_act_re) and not self._is_half_conv(mate)
                and (mate.attrs['kh'], mate.attrs['kw'], mate.attrs['rate']) == (1, 1, 1)
                and mate.inputs[0].shape[3] % 64 == 0)

    def _store_dt(self, result, in_dt):
        """Storage type of the tensor `result` a convolution reading `in_dt` tensors produces.
        Outside the mixed mode: the input's.  Mixed mode: half precision when every consumer is a
        half-precision convolution, or the residual add that such a convolution absorbs."""
        if not self.mixed:
            return in_dt
        if result.is_output or not result.consumers:
            return self.dt
        for c in result.consumers:
            if c.kind == 'conv2d' and c.inputs[0] is result and (self._is_half_conv(c) or self._takes_half_acts(c)):
                continue
            if c.kind == 'max_pool' and c.attrs['k'] > 1 and not c.is_output and self._store_dt(c, in_dt) == 'f16':
                continue        # the largest of half-precision numbers is one of them
            if c.kind == 'add' and len(c.inputs) == 2:
                # a residual: the half-precision convolution that absorbs the add reads it in the
                # type it stores its own result in
                other = c.inputs[0] if c.inputs[1] is result else c.inputs[1]
                if other is not result and self._is_half_conv(other) and other.pos > result.pos:
                    total = self._only_consumer(c, 'relu') or c
                    if self._store_dt(total, 'f16') == 'f16':
                        continue
            return self.dt
        return 'f16'

    def _dev_f32(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
        self.keepalive.append(t)
        return t

    def _val(self, ln):
        """Value of a node, materialising a deferred one on first real use."""
        if ln not in self.vals and ln in self.lazy:
            self.lazy.pop(ln)[-1]()
        return self.vals[ln]

    @staticmethod
    def _only_consumer(ln, kind):
        if ln.is_output or len(ln.consumers) != 1 or ln.consumers[0].kind != kind:
            return None
        return ln.consumers[0]

    def _emit(self, kind, name, fname, args, flops=0, nbytes=0, keep=None, graph_flops=None, mpp=1):
        st = Step(kind, name, getattr(lib, fname), args, flops, nbytes, keep, graph_flops)
        d = keep[0] if isinstance(keep, tuple) and keep else keep
        st.mpp = 3 if isinstance(d, _lib.ConvDesc) and d.in_dtype == _lib.F32_BF16X3 else mpp
        self.steps.append(st)
        self.step_fname.append(fname)

    # --------------------------------------------------------------- lowering
    def _lower(self):
        # concat buffers first: resize producers with one consumer write in place
        self.placement = {}
        for ln in self.order:
            if ln.kind != 'concat':
                continue
            buf = self._alloc(ln.shape, self.dt)
            self.vals[ln] = Val(buf, ln.shape, self.dt)
            off = 0
            for src in ln.inputs:
                if (src.kind == 'resize' and not src.is_output and len(src.consumers) == 1
                        and off % _EPC[self.dt] == 0):
                    self.placement[src] = (buf, off, ln.shape[3])
                off += src.shape[3]
        self.fused = set()
        handlers = {
            'placeholder': self._l_placeholder, 'pad_to_bbox': self._l_image_op,
            'to_float': self._l_image_op, 'affine': self._l_affine,
            'conv2d': self._l_conv, 'depthwise': self._l_depthwise,
            'max_pool': self._l_maxpool, 'avg_pool': self._l_avgpool,
            'resize': self._l_resize, 'add': self._l_add, 'relu': self._l_relu,
            'concat': self._l_concat, 'argmax': self._l_argmax, 'lut': self._l_lut,
            'crop': self._l_crop,
        }
        for ln in self.order:
            if ln in self.fused:
                continue
            handlers[ln.kind](ln)
        for ln in self.out_nodes.values():
            self._val(ln)

    # ---- input image chain
    def _l_placeholder(self, ln):
        if self.input_val is not None:
            raise NotImplementedError('a graph may have one placeholder')
        if ln.dtype != 'uint8' or ln.shape[3] != 3:
            raise NotImplementedError('placeholders must be uint8 images with 3 channels')
        buf = torch.zeros(ln.shape, dtype=torch.uint8, device=self.device)
        self.input_val = Val(buf, ln.shape, 'u8')
        self.vals[ln] = self.input_val
        self.image_src = {ln: ln}

    def _l_image_op(self, ln):
        src = self.image_src.get(ln.inputs[0])
        if src is None:
            raise NotImplementedError('%s is only supported on the input image' % ln.kind)
        self.image_src[ln] = src

    def _l_affine(self, ln):
        src = self.image_src.get(ln.inputs[0])
        if src is None:
            raise NotImplementedError('channel_affine is only supported on the input image')

        def materialise():
            n, ho, wo, c = ln.shape
            pc = _EPC[self.dt]
            out = Val(self._alloc(ln.shape, self.dt, phys_c=pc), ln.shape, self.dt, phys_c=pc)
            self.vals[ln] = out
            x = self.vals[src]
            _, h, w, _ = x.shape
            mean = (ctypes.c_float * 3)(*ln.attrs['mean'])
            scale = (ctypes.c_float * 3)(*ln.attrs['scale'])
            self._emit('preprocess', 'preprocess', 'fss_preprocess_u8',
                       (x.ptr, out.ptr, n, h, w, ho, wo, pc, mean, scale, _CODE[self.dt], self.sh),
                       nbytes=n * h * w * 3 + n * ho * wo * pc * _ESZ[self.dt], keep=(mean, scale))

        # deferred: a first convolution reads the uint8 frame itself (see _stem_source)
        self.lazy[ln] = ('preprocess', src, materialise)

    # ---- convolutions
    def _fold_bn(self, a, cout):
        W = self.weights
        if a['bn'] is not None and a['bn']['var'] not in W:
            # weights from a frozen graph whose batch norm is already folded: the kernel is
            # used as it is, with the constant vector the graph adds behind the convolution
            key = a['weights'].rsplit('/', 1)[0] + '/folded_bias'
            if key not in W:
                raise KeyError('neither batch norm statistics nor a folded bias for %s' % a['weights'])
            return np.ones(cout, np.float64), W[key].astype(np.float64)
        if a['bn'] is not None:
            bn = a['bn']
            scale = 1.0 / np.sqrt(W[bn['var']].astype(np.float64) + bn['epsilon'])
            if bn['gamma'] is not None:
                scale = scale * W[bn['gamma']].astype(np.float64)
            bias = -W[bn['mean']].astype(np.float64) * scale
            if bn['beta'] is not None:
                bias = bias + W[bn['beta']].astype(np.float64)
            return scale, bias
        return np.ones(cout, np.float64), W[a['biases']].astype(np.float64)

    def _stem_source(self, ln):
        """(preprocess node, sampling step, logical h, w) when conv `ln` can run as
        fss_stem_conv_u8: its input is the still-unmaterialised preprocessed frame, or
        an align_corners shrink of it by an exact integer factor (pure subsampling)."""
        a = ln.attrs
        inp, sub = ln.inputs[0], 1
        if not self.fuse_stem or inp in self.vals:
            return None
        d = self.lazy.get(inp)
        if d and d[0] == 'resize' and d[2]:
            small, big = inp.shape, d[1].shape
            if small[1] < 2 or small[2] < 2 or d[1] in self.vals:
                return None
            f = (big[1] - 1) // (small[1] - 1)
            if not (2 <= f <= 8 and (big[1] - 1) == f * (small[1] - 1) and (big[2] - 1) == f * (small[2] - 1)):
                return None
            inp, sub = d[1], f
            d = self.lazy.get(inp)
        if not d or d[0] != 'preprocess':
            return None
        if not (a['biases'] is None and ln.shape[3] <= 64 and a['kh'] * a['kw'] <= 64
                and a['stride'] <= 4 and a['rate'] <= 4 and self._only_consumer(ln, 'add') is None):
            return None
        return inp, sub, ln.inputs[0].shape[1], ln.inputs[0].shape[2]

    def _l_stem_conv(self, ln, pre, sub, h, w):
        a = ln.attrs
        x = self.vals[self.lazy[pre][1]]                      # the uint8 frames
        n, hr, wr, _ = x.shape
        _, ho, wo, cout = ln.shape
        kh, kw = a['kh'], a['kw']
        scale, bias = self._fold_bn(a, cout)
        Wt = self.weights[a['weights']].astype(np.float64) * scale      # [kh,kw,3,cout]
        Wt = np.pad(Wt, ((0, 0), (0, 0), (0, 1), (0, 0)))
        k_pad, cout_pad = ctypes.c_int(0), ctypes.c_int(0)
        nxt, dt2 = self._stem2_partner(ln)
        code = _CODE[dt2] if nxt is not None else self._in_code(self.dt)
        out_dt = self.dt
        if nxt is None and self.split and self._store_dt(ln, self.dt) == 'f16':
            out_dt, code = 'f16', _lib.F16_W2       # every reader takes half-precision activations
        _lib.call('fss_stem_pack_dims', kh, kw, cout, code, ctypes.byref(k_pad),
                  ctypes.byref(cout_pad))
        Wp = np.zeros((cout_pad.value, k_pad.value), np.float32)
        Wp[:cout, :kh * kw * 4] = Wt.transpose(3, 0, 1, 2).reshape(cout, -1)
        if nxt is not None:
            wdev = self._pack(Wp, dt2)
        elif self.dt == 'bf16':
            wdev = self._pack(Wp, 'bf16')
        elif self.split:
            wdev = torch.from_numpy(stem_pack_f16x2(Wt[:, :, :3], pre.attrs['mean'], pre.attrs['scale'],
                                                    cout_pad.value, k_pad.value).view(np.int16)).to(self.device)
            self.keepalive.append(wdev)
        else:
            wdev = self._pack(Wp, 'f32')
        bdev = self._dev_f32(bias)
        self.param_bytes += cout * kh * kw * 3 * _ESZ[self.dt]
        mean = (ctypes.c_float * 3)(*pre.attrs['mean'])
        sc = (ctypes.c_float * 3)(*pre.attrs['scale'])
        if nxt is not None:
            return self._l_stem2_conv(ln, nxt, x, sub, h, w, wdev, bdev, mean, sc, dt2)
        out = Val(self._alloc(ln.shape, out_dt), ln.shape, out_dt)
        self.vals[ln] = out
        self._emit('stem', a['weights'].rsplit('/', 1)[0], 'fss_stem_conv_u8',
                   (x.ptr, wdev.data_ptr(), bdev.data_ptr(), out.ptr, n, hr, wr, sub, h, w, mean, sc,
                    cout, kh, kw, a['stride'], a['rate'], a['pad_top'], a['pad_left'], ho, wo,
                    _lib.ACT_CODES[a['activation']], code, out.ld, self.sh),
                   flops=2 * n * ho * wo * cout * kh * kw * 3,
                   nbytes=n * (hr // sub) * (wr // sub) * 3 + n * ho * wo * cout * _ESZ[out_dt],
                   keep=(mean, sc), mpp=2 if code in (_lib.F32_BF16X3, _lib.F16_W2) else 1)

    def _stem2_partner(self, ln):
        """(the convolution that consumes stem conv `ln`, the 16-bit type) when both can run as
        one fss_stem2_conv_u8 launch (3x3, stride 2, batch norm, 32 or 64 channels; bfloat16, or
        half precision where the mixed mode puts both layers and their consumers), so that the
        output of `ln` never reaches memory; (None, None) otherwise."""
        def shape_ok(c):
            a = c.attrs
            return ((a['kh'], a['kw'], a['stride'], a['rate']) == (3, 3, 2, 1) and a['biases'] is None
                    and c.shape[3] in (32, 64) and a['pad_top'] <= 2 and a['pad_left'] <= 2)
        if not self.fuse_stem2 or not (self.dt == 'bf16' or self.mixed) or not shape_ok(ln):
            return None, None
        nxt = self._only_consumer(ln, 'conv2d')
        if nxt is None or not shape_ok(nxt) or self._only_consumer(nxt, 'add') is not None:
            return None, None
        if self.dt == 'bf16':
            return nxt, 'bf16'
        if self._is_half_conv(ln) and self._is_half_conv(nxt) and self._store_dt(nxt, 'f16') == 'f16':
            return nxt, 'f16'
        return None, None

    def _l_stem2_conv(self, ln, nxt, x, sub, h, w, w1dev, b1dev, mean, sc, dt):
        a1, a2 = ln.attrs, nxt.attrs
        n, hr, wr, _ = x.shape
        _, h1, w1, c1 = ln.shape
        _, h2, w2, c2 = nxt.shape
        scale, bias = self._fold_bn(a2, c2)
        Wt = self.weights[a2['weights']].astype(np.float64) * scale      # [3,3,c1,c2]
        k_pad, cout_pad = _lib.conv_pack_dims(c1, 3, 3, c2, _CODE[dt])
        Wp = np.zeros((cout_pad, k_pad), np.float32)
        Wp[:c2, :9 * c1] = Wt.transpose(3, 0, 1, 2).reshape(c2, -1)
        w2dev = self._pack(Wp, dt)
        b2dev = self._dev_f32(bias)
        self.param_bytes += c2 * 9 * c1 * _ESZ[dt]
        out = Val(self._alloc(nxt.shape, dt), nxt.shape, dt)
        self.vals[nxt] = out
        self.fused.add(nxt)
        self._emit('stem2', a2['weights'].rsplit('/', 1)[0], 'fss_stem2_conv_u8',
                   (x.ptr, w1dev.data_ptr(), b1dev.data_ptr(), w2dev.data_ptr(), b2dev.data_ptr(), out.ptr,
                    n, hr, wr, sub, h, w, mean, sc,
                    c1, a1['pad_top'], a1['pad_left'], h1, w1, _lib.ACT_CODES[a1['activation']],
                    c2, a2['pad_top'], a2['pad_left'], h2, w2, _lib.ACT_CODES[a2['activation']],
                    _CODE[dt], out.ld, self.sh),
                   flops=2 * n * (h1 * w1 * c1 * 27 + h2 * w2 * c2 * 9 * c1),
                   nbytes=n * (hr // sub) * (wr // sub) * 3 + n * h2 * w2 * c2 * _ESZ[dt],
                   keep=(mean, sc))

    def _dual_partner(self, ln):
        """The 1x1 convolution A when `relu?(conv ln + A)` can run as one
        fss_conv2d_dual_nhwc launch (the cascade feature fusion; a bottleneck's projection
        shortcut + its last 1x1 conv): both without bias and activation, stride 1, the add
        their only consumer, A later in the order (the launch is emitted at A's position,
        where both inputs exist), channel counts multiples of 64."""
        a = ln.attrs
        if not self.fuse_dual or a['biases'] is not None or a['activation'] is not None or a['stride'] != 1:
            return None
        add = self._only_consumer(ln, 'add')
        if add is None or len(add.inputs) != 2:
            return None
        A = add.inputs[0] if add.inputs[1] is ln else add.inputs[1]
        if A is ln or A.kind != 'conv2d' or A.pos < ln.pos or self._only_consumer(A, 'add') is not add:
            return None
        b = A.attrs
        if ((b['kh'], b['kw'], b['stride'], b['rate']) != (1, 1, 1, 1) or b['biases'] is not None
                or b['activation'] is not None or A.shape != ln.shape):
            return None
        if ln.inputs[0].shape[3] % 64 or A.inputs[0].shape[3] % 64:
            return None
        if self._is_half_conv(ln) != self._is_half_conv(A):
            return None        # mixed mode: the two inputs would differ in type
        return A

    def _chain_partner(self, result, k, cout, dt, parts=None):
        """The 1x1 convolution (batch norm, no biases, stride 1, 32 or 64 filters) among the
        consumers of `result` that can run in the same launch as the 1x1 convolution producing
        `result` (fss_conv1x1_chain_nhwc: the end of one bottleneck unit and the first layer of
        the next, where both are bound by the bytes of `result`)."""
        if (not self.fuse_chain or dt != 'bf16' or k not in (32, 64, 128) or cout not in (128, 256)
                or result.is_output or any(p % 32 for p in (parts or (k,)))):
            return None
        for c in result.consumers:
            if c.kind != 'conv2d' or c.inputs[0] is not result or c in self.fused or c in self.dual:
                continue
            a = c.attrs
            if ((a['kh'], a['kw'], a['stride'], a['rate']) != (1, 1, 1, 1) or a['biases'] is not None
                    or c.shape[3] not in (32, 64) or c.shape[:3] != result.shape[:3]):
                continue
            if self._dual_partner(c) is not None or self._only_consumer(c, 'add') is not None:
                continue
            return c
        return None

    def _emit_chain(self, nxt, nodes, x, x2, Wm, bias, res, act, name):
        """Emits conv1x1([x | x2], Wm) + bias + res -> act (the value of `nodes`) and the 1x1
        convolution `nxt` of that value as one launch."""
        n, ho, wo, cout = nodes[-1].shape
        cout2 = nxt.shape[3]
        cin, cin2 = x.shape[3], (x2.shape[3] if x2 is not None else 0)
        k = cin + cin2
        wdev = torch.from_numpy(f32_to_bf16_bits(np.ascontiguousarray(Wm, np.float32)).view(np.int16)).to(self.device)
        s2, b2 = self._fold_bn(nxt.attrs, cout2)
        W2 = (self.weights[nxt.attrs['weights']].astype(np.float64) * s2).reshape(cout, cout2).T
        w2dev = torch.from_numpy(f32_to_bf16_bits(np.ascontiguousarray(W2, np.float32)).view(np.int16)).to(self.device)
        self.keepalive.extend([wdev, w2dev])
        bdev, b2dev = self._dev_f32(bias), self._dev_f32(b2)
        self.param_bytes += (cout * k + cout2 * cout) * 2
        out = Val(self._alloc(nodes[-1].shape, 'bf16'), nodes[-1].shape, 'bf16')
        out2 = Val(self._alloc(nxt.shape, 'bf16'), nxt.shape, 'bf16')
        for node in nodes:
            self.vals[node] = out
        self.vals[nxt] = out2
        self.fused.add(nxt)
        m = n * ho * wo
        self._emit('chain', name + '+' + '/'.join(nxt.attrs['weights'].split('/')[-4:-1:2]),
                   'fss_conv1x1_chain_nhwc',
                   (x.ptr, cin, x.ld, x2.ptr if x2 is not None else None, cin2, x2.ld if x2 is not None else 0,
                    wdev.data_ptr(), bdev.data_ptr(), res.ptr if res is not None else None,
                    res.ld if res is not None else 0, _lib.ACT_CODES[act], out.ptr, out.ld, cout,
                    w2dev.data_ptr(), b2dev.data_ptr(), _lib.ACT_CODES[nxt.attrs['activation']], out2.ptr,
                    out2.ld, cout2, m, _CODE['bf16'], self.sh),
                   flops=2 * m * cout * (k + cout2),
                   nbytes=(m * (k + cout * (2 if res is not None else 1) + cout2) + cout * (k + cout2)) * 2)

    def _head_partner(self, result, cout, dt, wide=False):
        """The 1x1 classifier (a convolution with biases, no normalisation or activation, at
        most 20 classes) that is the only consumer of `result`, when it can run in the
        epilogue of the convolution producing `result` (fss_conv2d_dual_head_nhwc)."""
        if not self.fuse_head or self.mfma_edge != 16 or cout % 32 or cout > 256:
            return None
        if dt not in ('bf16', 'f32s') and not (wide and dt == 'f16'):
            return None
        c = self._only_consumer(result, 'conv2d')
        if c is None or c in self.dual or self._dual_partner(c) is not None:
            return None
        a = c.attrs
        if ((a['kh'], a['kw'], a['stride'], a['rate']) != (1, 1, 1, 1) or a['biases'] is None
                or a['activation'] is not None or c.shape[3] > 20 or c.shape[:3] != result.shape[:3]):
            return None
        return c

    def _emit_head(self, head, d, x, x2, cin2, wdev, bdev, cout, k, name, res=None, zres=None, graph_k=None):
        """Emits conv (+ second input) + classifier `head` as one launch; the activated
        cout-channel tensor gets no buffer."""
        n, ho, wo, ncls = head.shape
        sc, bh = self._fold_bn(head.attrs, ncls)
        Wh = np.zeros((32, cout), np.float32)
        Wh[:ncls] = (self.weights[head.attrs['weights']].astype(np.float64) * sc).reshape(cout, ncls).T
        esz = 2 if x.dt == 'bf16' else 4      # bytes per element of the residuals and their sources
        if x.dt == 'bf16':
            wh_bits = f32_to_bf16_bits(Wh)
        else:
            # float32x3: bf16 high parts of the float weights, then the bf16 remainders
            hi = f32_to_bf16_bits(Wh)
            hi_f = (hi.astype(np.uint32) << 16).view(np.float32)
            wh_bits = np.concatenate([hi, f32_to_bf16_bits(Wh - hi_f)], axis=0)
        whdev = torch.from_numpy(np.ascontiguousarray(wh_bits).view(np.int16)).to(self.device)
        self.keepalive.append(whdev)
        bhdev = self._dev_f32(bh)
        self.param_bytes += cout * ncls * esz
        pc_out = (ncls + 3) // 4 * 4       # as the unfused classifier: float4-readable pixels
        out = Val(torch.zeros((n, ho, wo, pc_out), dtype=torch.float32, device=self.device),
                  head.shape, 'f32', phys_c=pc_out)
        self.vals[head] = out
        self.fused.add(head)
        nbytes = ((x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3] + n * ho * wo * cin2) * _ESZ[x.dt]
                  + (cout * k + cout * ncls + (n * ho * wo * cout if res is not None else 0)
                     + (sum(int(np.prod(v.shape)) for v in zres) if zres else 0)) * esz
                  + n * ho * wo * ncls * 4)
        zptrs = (ctypes.c_void_p * 4)(*[v.ptr for v in zres]) if zres else None
        self._emit('conv', name + '+' + head.attrs['weights'].rsplit('/', 2)[-2],
                   'fss_conv2d_dual_head_nhwc',
                   (ctypes.byref(d), x.ptr, x2.ptr if x2 is not None else None, cin2, wdev.data_ptr(),
                    bdev.data_ptr(), res.ptr if res is not None else None, zptrs,
                    zres[0].shape[1] if zres else 0, zres[0].shape[2] if zres else 0, whdev.data_ptr(),
                    bhdev.data_ptr(), ncls, out.ptr, out.ld, self.sh),
                   flops=2 * n * ho * wo * cout * (k + ncls), nbytes=nbytes, keep=(d, zptrs),
                   graph_flops=2 * n * ho * wo * cout * ((graph_k or k) + ncls),
                   mpp=2 if d.in_dtype == _lib.F16_W2 else 1)

    def _zoom_source(self, B):
        """The low-resolution node x when conv B is conv3x3, dilation 2, stride 1, SAME of a
        deferred align_corners zoom of x to (2h-1, 2w-1) that nothing else reads (the dilated
        convolution of a cascade feature fusion, architectures/icnet_architecture.py:238-247)."""
        # None: only from FSS_COMMUTE_MIN_BATCH frames per launch on (the six extra launches
        # cost more than the saved FLOP at small batches)
        if self.commute_zoom is False:
            return None
        if self.commute_zoom is None and B.shape[0] < int(os.environ.get('FSS_COMMUTE_MIN_BATCH', '8')):
            return None
        b, src = B.attrs, B.inputs[0]
        d = self.lazy.get(src)
        if d is None and src.kind == 'resize' and src not in self.vals:
            # asked before the zoom has been lowered (a decision about an earlier tensor that
            # depends on it): every resize is deferred when its turn comes
            d = ('resize', src.inputs[0], src.attrs['align_corners'])
        if not d or d[0] != 'resize' or not d[2] or src.is_output or len(src.consumers) != 1:
            return None
        if (b['kh'], b['kw'], b['stride'], b['rate'], b['pad_top'], b['pad_left']) != (3, 3, 1, 2, 2, 2):
            return None
        low = d[1]
        _, h, w, c = low.shape
        if h < 4 or w < 4 or tuple(src.shape[1:3]) != (2 * h - 1, 2 * w - 1) or B.shape[:3] != src.shape[:3]:
            return None
        return low

    def _l_conv_dual_commuted(self, A, B, xl, x2):
        """relu?(conv3x3_d2(zoom_x2(xl)) + conv1x1(x2)) with the 3x3 computed at xl's resolution:
        zoom and convolution are both linear, so conv3x3_d2(zoom(xl)) = zoom(conv3x3(xl)) except
        on the output rows / columns next to the border ring, which fss_zoom2_conv_commute
        takes from convolutions of border crops of xl (see include/fss.h).  The 1x1 launch then
        adds the zoomed result as its residual."""
        dt = xl.dt
        b = B.attrs
        n, h, w, cin = xl.shape
        _, ho, wo, cout = B.shape
        cin2 = x2.shape[3]
        add = A.consumers[0]
        act, result = None, add
        self.fused.add(add)
        relu = self._only_consumer(add, 'relu')
        if relu is not None:
            act, result = 'relu', relu
            self.fused.add(relu)
        sB, bB = self._fold_bn(b, cout)
        sA, bA = self._fold_bn(A.attrs, cout)

        def pack(Wt, k, form=dt):
            k_pad, cout_pad = (k + 63) // 64 * 64, (cout + 255) // 256 * 256
            Wp = np.zeros((cout_pad, k_pad), np.float32)
            Wp[:cout, :k] = Wt
            return self._pack(Wp, form), k_pad, cout_pad

        WB = (self.weights[b['weights']].astype(np.float64) * sB).transpose(3, 0, 1, 2).reshape(cout, -1)
        wB, k_pad, cout_pad = pack(WB, 9 * cin)
        self.param_bytes += cout * (9 * cin + cin2) * _ESZ[dt]
        base = b['weights'].rsplit('/', 1)[0]
        esz = _ESZ[dt]
        tb = Val(self._alloc((2 * n, 3, w, cin), dt), (2 * n, 3, w, cin), dt)
        lr = Val(self._alloc((2 * n, h, 3, cin), dt), (2 * n, h, 3, cin), dt)
        cn = Val(self._alloc((4 * n, 3, 3, cin), dt), (4 * n, 3, 3, cin), dt)
        crop_px = 2 * n * 3 * w + 2 * n * h * 3 + 4 * n * 9
        self._emit('copy', base + '/border_crops', 'fss_zoom2_border_gather',
                   (xl.ptr, tb.ptr, lr.ptr, cn.ptr, n, h, w, cin, _CODE[dt], self.sh),
                   nbytes=2 * crop_px * cin * esz)
        head = self._head_partner(result, cout, dt) if cin2 % 64 == 0 else None
        # results only the classifier launch reads (it interpolates them itself) stay plain
        # float in the split mode: nothing multiplies them, so splitting them would be wasted
        ydt = 'f32' if (dt == 'f32s' and head is not None and self.fuse_zoom_residual) else dt
        ys = []
        # the crops are convolved without padding across their 3 rows / columns: only the
        # middle row / column / pixel comes out
        for tag, v, pt, pl in (('@source', xl, 1, 1), ('@rows', tb, 0, 1), ('@columns', lr, 1, 0),
                               ('@corners', cn, 0, 0)):
            nn, hh, ww, _ = v.shape
            oh, ow = (hh if pt else 1), (ww if pl else 1)
            y = Val(self._alloc((nn, oh, ow, cout), ydt), (nn, oh, ow, cout), ydt)
            d = _lib.ConvDesc(n=nn, h=hh, w=ww, cin=cin, cout=cout, kh=3, kw=3, stride=1, dilation=1,
                              pad_top=pt, pad_left=pl, ho=oh, wo=ow, act=_lib.ACT_CODES[None],
                              in_dtype=self._in_code(dt), out_dtype=_CODE[ydt], y_ld=y.ld, res_ld=0,
                              k_pad=k_pad, cout_pad=cout_pad)
            self._emit('conv', base + tag, 'fss_conv2d_nhwc',
                       (ctypes.byref(d), v.ptr, wB.data_ptr(), None, None, y.ptr, self.sh),
                       flops=2 * nn * oh * ow * cout * 9 * cin,
                       nbytes=(nn * hh * ww * cin + nn * oh * ow * cout + cout * 9 * cin) * esz, keep=d,
                       graph_flops=0)
            ys.append(y)
        self.temps.extend([tb, lr, cn] + ys)        # buffers no graph node owns
        r = None
        if head is None or not self.fuse_zoom_residual:
            r = Val(self._alloc((n, ho, wo, cout), dt), (n, ho, wo, cout), dt)
            self._emit('resize', base + '/zoom', 'fss_zoom2_conv_commute',
                       (ys[0].ptr, ys[1].ptr, ys[2].ptr, ys[3].ptr, r.ptr, n, h, w, cout, _CODE[dt], self.sh),
                       nbytes=(n * h * w + n * ho * wo) * cout * esz)
            self.temps.append(r)
        # the 1x1 half, with the zoomed 3x3 half as its residual
        # (a half-precision x2: read against two-f16 weights, FSS_F16_W2)
        wide = x2.dt == 'f16'
        WA = (self.weights[A.attrs['weights']].astype(np.float64) * sA).reshape(cin2, cout).T
        wA, k_pad2, _ = pack(WA, cin2, 'f16w' if wide else dt)
        bdev = self._dev_f32(bA + bB)
        name = base + '+' + A.attrs['weights'].rsplit('/', 2)[-2]
        d = _lib.ConvDesc(n=n, h=ho, w=wo, cin=cin2, cout=cout, kh=1, kw=1, stride=1, dilation=1,
                          pad_top=0, pad_left=0, ho=ho, wo=wo, act=_lib.ACT_CODES[act],
                          in_dtype=_lib.F16_W2 if wide else self._in_code(dt), out_dtype=_CODE[dt], y_ld=cout,
                          res_ld=r.ld if r else 0, k_pad=k_pad2, cout_pad=cout_pad)
        if head is not None:
            # the classifier launch interpolates the source-resolution results itself
            return self._emit_head(head, d, x2, None, 0, wA, bdev, cout, cin2, name, res=r,
                                   zres=None if r is not None else ys, graph_k=9 * cin + cin2)
        out = Val(self._alloc(B.shape, dt), B.shape, dt)
        for node in (A, B, add, result):
            self.vals[node] = out
        d.y_ld = out.ld
        self._emit('conv', name, 'fss_conv2d_nhwc',
                   (ctypes.byref(d), x2.ptr, wA.data_ptr(), bdev.data_ptr(), r.ptr, out.ptr, self.sh),
                   flops=2 * n * ho * wo * cout * cin2,
                   nbytes=(n * ho * wo * 2 * cout + cout * cin2) * esz + n * ho * wo * cin2 * _ESZ[x2.dt], keep=d,
                   graph_flops=2 * n * ho * wo * cout * (9 * cin + cin2), mpp=2 if wide else 1)

    def _l_conv_dual(self, A, B):
        """relu?(conv B + conv1x1 A) in one launch; A is the node being lowered."""
        low = self._zoom_source(B)
        if low is not None:
            xl, x2 = self._val(low), self._val(A.inputs[0])
            half2 = x2.dt == 'f16' and self._takes_half_acts(A)     # read with two-f16 weights
            if (xl.dense and x2.dense and xl.dt == self.dt and (x2.dt == self.dt or half2) and xl.phys_c == low.shape[3]
                    and x2.phys_c == A.inputs[0].shape[3] and xl.phys_c % _EPC[xl.dt] == 0
                    and B.shape[3] % _EPC[xl.dt] == 0):
                self.lazy.pop(B.inputs[0])
                return self._l_conv_dual_commuted(A, B, xl, x2)
        x, x2 = self._val(B.inputs[0]), self._val(A.inputs[0])
        if not (x.dense and x2.dense and x.dt == x2.dt and x.phys_c % 64 == 0 and x2.phys_c % 64 == 0
                and x.phys_c == B.inputs[0].shape[3] and x2.phys_c == A.inputs[0].shape[3]):
            raise NotImplementedError('two-input conv needs dense inputs with channels in multiples of 64')
        b = B.attrs
        n, h, w, cin = x.shape
        cin2 = x2.shape[3]
        _, ho, wo, cout = B.shape
        kh, kw = b['kh'], b['kw']
        add = A.consumers[0]
        act, result = None, add
        self.fused.add(add)
        relu = self._only_consumer(add, 'relu')
        if relu is not None:
            act, result = 'relu', relu
            self.fused.add(relu)
        sB, bB = self._fold_bn(b, cout)
        sA, bA = self._fold_bn(A.attrs, cout)
        WB = self.weights[b['weights']].astype(np.float64) * sB          # [kh,kw,cin,cout]
        WA = self.weights[A.attrs['weights']].astype(np.float64) * sA    # [1,1,cin2,cout]
        k = kh * kw * cin + cin2
        k_pad, cout_pad = (k + 63) // 64 * 64, (cout + 255) // 256 * 256
        Wp = np.zeros((cout_pad, k_pad), np.float32)
        Wp[:cout, :kh * kw * cin] = WB.transpose(3, 0, 1, 2).reshape(cout, -1)
        Wp[:cout, kh * kw * cin:k] = WA.reshape(cin2, cout).T
        # half-precision inputs of convolutions that keep float-quality weights (FSS_F16_W2)
        wide = x.dt == 'f16' and self._takes_half_acts(A) and self._takes_half_acts(B)
        wdev = self._pack(Wp, 'f16w' if wide else x.dt)
        bdev = self._dev_f32(bA + bB)
        self.param_bytes += cout * k * _ESZ[x.dt]
        name = b['weights'].rsplit('/', 1)[0] + '+' + A.attrs['weights'].rsplit('/', 2)[-2]
        if (kh, kw, b['rate'], b['pad_top'], b['pad_left']) == (1, 1, 1, 0, 0) and (h, w) == (ho, wo):
            nxt = self._chain_partner(result, k, cout, x.dt, parts=(cin, cin2))
            if nxt is not None:
                return self._emit_chain(nxt, (A, B, add, result), x, x2, Wp[:cout, :k], bA + bB, None, act, name)
        head = self._head_partner(result, cout, x.dt)
        if head is not None:
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=kh, kw=kw, stride=1,
                              dilation=b['rate'], pad_top=b['pad_top'], pad_left=b['pad_left'],
                              ho=ho, wo=wo, act=_lib.ACT_CODES[act], in_dtype=self._in_code(x.dt),
                              out_dtype=_CODE[x.dt], y_ld=cout, res_ld=0, k_pad=k_pad, cout_pad=cout_pad)
            return self._emit_head(head, d, x, x2, cin2, wdev, bdev, cout, k, name)
        out_dt = self._store_dt(result, x.dt)
        out = Val(self._alloc(B.shape, out_dt), B.shape, out_dt)
        for node in (A, B, add, result):
            self.vals[node] = out
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=kh, kw=kw, stride=1,
                          dilation=b['rate'], pad_top=b['pad_top'], pad_left=b['pad_left'],
                          ho=ho, wo=wo, act=_lib.ACT_CODES[act],
                          in_dtype=_lib.F16_W2 if wide else self._in_code(x.dt),
                          out_dtype=_CODE[out_dt], y_ld=out.ld, res_ld=0, k_pad=k_pad, cout_pad=cout_pad)
        nbytes = ((n * h * w * cin + n * ho * wo * cin2 + cout * k * (2 if wide else 1)) * _ESZ[x.dt]
                  + n * ho * wo * cout * _ESZ[out_dt])
        self._emit('conv', name, 'fss_conv2d_dual_nhwc',
                   (ctypes.byref(d), x.ptr, x2.ptr, cin2, wdev.data_ptr(), bdev.data_ptr(), None,
                    out.ptr, self.sh),
                   flops=2 * n * ho * wo * cout * k, nbytes=nbytes, keep=d, mpp=2 if wide else 1)

    def _pyramid_source(self, ln):
        """(base node, level values) when conv `ln` is the 1x1 convolution behind the deferred sum
        x + sum_f resize(pool_f(x)) of a fast pyramid pooling module
        (architectures/icnet_architecture.py:194-210) that nothing else reads."""
        # None: only from FSS_COMMUTE_MIN_BATCH frames per launch on (the extra small launches
        # cost more than the saved pass over the tensor at small batches)
        if self.commute_pyramid is False or os.environ.get('FSS_COMMUTE_PYRAMID', '1') == '0':
            return None
        if self.commute_pyramid is None and ln.shape[0] < int(os.environ.get('FSS_COMMUTE_MIN_BATCH', '8')):
            return None
        a, src = ln.attrs, ln.inputs[0]
        d = self.lazy.get(src)
        if not d or d[0] != 'pyramid' or src.is_output or len(src.consumers) != 1:
            return None
        if ((a['kh'], a['kw'], a['stride'], a['rate'], a['pad_top'], a['pad_left']) != (1, 1, 1, 1, 0, 0)
                or a['biases'] is not None or self._only_consumer(ln, 'add') is not None
                or ln in self.dual or self._dual_partner(ln) is not None):
            return None
        if ln.shape[3] % _EPC[self.dt] or d[1].shape[3] % 64:
            return None
        return d[1], d[2]

    def _l_conv_pyramid(self, ln, base, levels):
        """conv1x1(x + sum_f resize(p_f)) as conv1x1(x) + sum_f resize(conv1x1(p_f)): a 1x1
        convolution acts on every pixel alone and is linear, the align_corners resize is a
        weighted mean of pixels, so the two commute.  The small maps are convolved first (a few
        hundred pixels each), their resized sum is formed at the OUTPUT channel count and enters
        the convolution of x as its residual; the sum at the input channel count (a read and a
        write of the whole tensor) is never formed."""
        a = ln.attrs
        x = self._val(base)
        if not (x.dense and x.phys_c == base.shape[3]):
            raise NotImplementedError('pyramid base must be dense')
        self.lazy.pop(ln.inputs[0])
        dt = x.dt
        n, h, w, cin = x.shape
        cout = ln.shape[3]
        scale, bias = self._fold_bn(a, cout)
        Wt = self.weights[a['weights']].astype(np.float64) * scale
        k_pad, cout_pad = _lib.conv_pack_dims(cin, 1, 1, cout, _CODE[dt])
        Wp = np.zeros((cout_pad, k_pad), np.float32)
        Wp[:cout, :cin] = Wt.reshape(cin, cout).T
        wdev = self._pack(Wp, dt)
        bdev = self._dev_f32(bias)
        self.param_bytes += cout * cin * _ESZ[dt]
        name = a['weights'].rsplit('/', 1)[0]
        esz = _ESZ[dt]
        small = []
        for i, t in enumerate(levels):
            nn, bh, bw, _ = t.shape
            y = Val(self._alloc((nn, bh, bw, cout), dt), (nn, bh, bw, cout), dt)
            d = _lib.ConvDesc(n=nn, h=bh, w=bw, cin=cin, cout=cout, kh=1, kw=1, stride=1, dilation=1,
                              pad_top=0, pad_left=0, ho=bh, wo=bw, act=_lib.ACT_CODES[None],
                              in_dtype=self._in_code(dt), out_dtype=_CODE[dt], y_ld=y.ld, res_ld=0,
                              k_pad=k_pad, cout_pad=cout_pad)
            self._emit('conv', '%s@level%d' % (name, i), 'fss_conv2d_nhwc',
                       (ctypes.byref(d), t.ptr, wdev.data_ptr(), None, None, y.ptr, self.sh),
                       flops=2 * nn * bh * bw * cout * cin,
                       nbytes=(nn * bh * bw * (cin + cout) + cout * cin) * esz, keep=d, graph_flops=0)
            small.append(y)
        r = Val(self._alloc((n, h, w, cout), dt), (n, h, w, cout), dt)
        L = len(small)
        ptrs = (ctypes.c_void_p * L)(*[t.ptr for t in small])
        bhs = (ctypes.c_int * L)(*[t.shape[1] for t in small])
        bws = (ctypes.c_int * L)(*[t.shape[2] for t in small])
        self._emit('pyramid', name + '/pyramid_sum', 'fss_pyramid_add_nhwc',
                   (None, ptrs, bhs, bws, L, r.ptr, n, h, w, cout, _CODE[dt], self.sh),
                   nbytes=n * h * w * cout * esz, keep=(ptrs, bhs, bws))
        self.temps.extend(small + [r])
        out = Val(self._alloc(ln.shape, dt), ln.shape, dt)
        self.vals[ln] = out
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=1, kw=1, stride=1, dilation=1,
                          pad_top=0, pad_left=0, ho=h, wo=w, act=_lib.ACT_CODES[a['activation']],
                          in_dtype=self._in_code(dt), out_dtype=_CODE[dt], y_ld=out.ld, res_ld=r.ld,
                          k_pad=k_pad, cout_pad=cout_pad)
        self._emit('conv', name, 'fss_conv2d_nhwc',
                   (ctypes.byref(d), x.ptr, wdev.data_ptr(), bdev.data_ptr(), r.ptr, out.ptr, self.sh),
                   flops=2 * n * h * w * cout * cin,
                   nbytes=(n * h * w * (cin + 2 * cout) + cout * cin) * esz, keep=d)

    def _l_conv(self, ln):
        a = ln.attrs
        stem = self._stem_source(ln)
        if stem is not None:
            return self._l_stem_conv(ln, *stem)
        pyr = self._pyramid_source(ln)
        if pyr is not None:
            return self._l_conv_pyramid(ln, *pyr)
        if ln in self.dual:
            return self._l_conv_dual(ln, self.dual[ln])
        partner = self._dual_partner(ln)
        if partner is not None:
            self.dual[partner] = ln        # emitted when the 1x1 partner is lowered
            return
        x = self._val(ln.inputs[0])
        if not x.dense:
            raise NotImplementedError('conv input must be dense')
        n, h, w, cin = x.shape
        _, ho, wo, cout = ln.shape
        kh, kw = a['kh'], a['kw']
        in_dt = x.dt
        act = a['activation']
        res, result = None, ln
        add = self._only_consumer(ln, 'add')
        if add is not None and act is None and a['biases'] is None:
            other = [y for y in add.inputs if y is not ln]
            other = other[0] if other else None
            if other is not None and other.pos < ln.pos:
                ov = self._val(other)
                total = self._only_consumer(add, 'relu') or add
                # the residual's type: the input's; for half-precision input, the output's
                want = self._store_dt(total, in_dt) if in_dt == 'f16' else in_dt
                if ov.dt == want:
                    res, result = ov, add
                    self.fused.add(add)
                    if total is not add:
                        act, result = 'relu', total
                        self.fused.add(total)
        # logits stay float32; otherwise the input's type (mixed mode: what the consumers read)
        out_dt = 'f32' if a['biases'] is not None else self._store_dt(result, in_dt)
        scale, bias = self._fold_bn(a, cout)
        Wt = self.weights[a['weights']].astype(np.float64) * scale      # [kh,kw,cin,cout]
        pc = x.phys_c
        if pc != cin:
            Wt = np.pad(Wt, ((0, 0), (0, 0), (0, pc - cin), (0, 0)))
        k_pad, cout_pad = _lib.conv_pack_dims(pc, kh, kw, cout, _CODE[in_dt])
        Wp = np.zeros((cout_pad, k_pad), np.float32)
        Wp[:cout, :kh * kw * pc] = Wt.transpose(3, 0, 1, 2).reshape(cout, -1)
        wdev = self._pack(Wp, in_dt)
        bdev = self._dev_f32(bias)
        self.param_bytes += cout * kh * kw * cin * _ESZ[in_dt]
        if ((kh, kw, a['stride'], a['rate'], a['pad_top'], a['pad_left']) == (1, 1, 1, 1, 0, 0) and pc == cin
                and a['biases'] is None and out_dt == in_dt and (h, w) == (ho, wo)
                and (res is None or (res.dense and res.phys_c == cout))):
            nxt = self._chain_partner(result, cin, cout, in_dt)
            if nxt is not None:
                nodes = (ln,) if result is ln else ((ln, add, result) if result is not add else (ln, add))
                return self._emit_chain(nxt, nodes, x, None, Wp[:cout, :cin], bias, res, act,
                                        a['weights'].rsplit('/', 1)[0])
        if (self.resident_3x3 and (kh, kw) == (3, 3) and pc == cin == 64 and cout in (64, 128)
                and in_dt in ('bf16', 'f16') and out_dt == in_dt and res is None and a['biases'] is None
                and k_pad == 9 * 64 and a['stride'] <= 4 and a['rate'] <= 8 and x.ld % 8 == 0
                and (h + 2 * a['rate'] + a['pad_top']) * w * x.ld < 2 ** 31 - 1 and ho * wo < 2 ** 31 - 1
                and (cout == 128 or a['stride'] == 2 or (a['stride'], a['rate']) == (1, 1))):
            # 64 input channels: the filter stays in LDS, the activations go straight to the MFMAs
            # (measured faster than the implicit GEMM for 128 filters and for stride 2: `Conv_2`
            # 0.92 -> 0.80 ms, block1/unit_3/conv2 0.212 -> 0.202 ms per 128 frames; and at stride 1,
            # dilation 1, where the horizontal taps are shared between lanes)
            out = Val(self._alloc(ln.shape, out_dt), ln.shape, out_dt)
            self.vals[result] = out
            self._emit('conv3x3', a['weights'].rsplit('/', 1)[0], 'fss_conv3x3_c64_nhwc',
                       (x.ptr, x.ld, wdev.data_ptr(), bdev.data_ptr(), out.ptr, out.ld, n, h, w, ho, wo, cout,
                        a['stride'], a['rate'], a['pad_top'], a['pad_left'], _lib.ACT_CODES[act], _CODE[in_dt],
                        self.sh),
                       flops=2 * n * ho * wo * cout * 9 * cin,
                       nbytes=(n * h * w * cin + n * ho * wo * cout + cout * 9 * cin) * 2)
            return
        if in_dt == 'f16' and pc == cin and self._takes_half_acts(ln):
            # half-precision activations, float-quality weights (FSS_F16_W2): per output channel
            # the k_pad high parts f16(W), then the k_pad remainders f16(W - high)
            out = Val(self._alloc(ln.shape, out_dt), ln.shape, out_dt)
            self.vals[result] = out
            if result is not ln:
                self.vals[ln] = out
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=1, kw=1, stride=1, dilation=1, pad_top=0,
                              pad_left=0, ho=ho, wo=wo, act=_lib.ACT_CODES[act], in_dtype=_lib.F16_W2,
                              out_dtype=_CODE[out_dt], y_ld=out.ld, res_ld=res.ld if res else 0,
                              k_pad=k_pad, cout_pad=cout_pad)
            nbytes = ((n * h * w * cin + 2 * cout * cin) * 2 + n * ho * wo * cout * _ESZ[out_dt]
                      + (n * ho * wo * cout * _ESZ[res.dt] if res else 0))
            self._emit('conv', a['weights'].rsplit('/', 1)[0], 'fss_conv2d_nhwc',
                       (ctypes.byref(d), x.ptr, self._pack(Wp, 'f16w').data_ptr(), bdev.data_ptr(),
                        res.ptr if res else None, out.ptr, self.sh),
                       flops=2 * n * ho * wo * cout * cin, nbytes=nbytes, keep=d, mpp=2)
            return
        head = None
        if res is None and a['biases'] is None and pc == cin and cin % 64 == 0 and a['stride'] == 1:
            head = self._head_partner(result, cout, in_dt)
        if head is not None:
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=pc, cout=cout, kh=kh, kw=kw, stride=1,
                              dilation=a['rate'], pad_top=a['pad_top'], pad_left=a['pad_left'],
                              ho=ho, wo=wo, act=_lib.ACT_CODES[act], in_dtype=self._in_code(in_dt),
                              out_dtype=_CODE[in_dt], y_ld=cout, res_ld=0, k_pad=k_pad, cout_pad=cout_pad)
            return self._emit_head(head, d, x, None, 0, wdev, bdev, cout, kh * kw * cin,
                                   a['weights'].rsplit('/', 1)[0])
        if a['biases'] is not None:
            # logits: pixel stride rounded up to 16 bytes so later kernels read float4s;
            # the pad channel is zero and never looked at
            pc_out = (cout + 3) // 4 * 4
            out = Val(torch.zeros((n, ho, wo, pc_out), dtype=torch.float32, device=self.device),
                      ln.shape, out_dt, phys_c=pc_out)
        else:
            out = Val(self._alloc(ln.shape, out_dt), ln.shape, out_dt)
        self.vals[result] = out
        if result is not ln:
            self.vals[ln] = out
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=pc, cout=cout, kh=kh, kw=kw, stride=a['stride'],
                          dilation=a['rate'], pad_top=a['pad_top'], pad_left=a['pad_left'],
                          ho=ho, wo=wo, act=_lib.ACT_CODES[act], in_dtype=self._in_code(in_dt),
                          out_dtype=_CODE[out_dt], y_ld=out.ld, res_ld=res.ld if res else 0,
                          k_pad=k_pad, cout_pad=cout_pad)
        nbytes = ((n * h * w * pc + cout * kh * kw * pc) * _ESZ[in_dt]
                  + n * ho * wo * cout * _ESZ[out_dt]
                  + (n * ho * wo * cout * _ESZ[in_dt] if res else 0))
        self._emit('conv', a['weights'].rsplit('/', 1)[0], 'fss_conv2d_nhwc',
                   (ctypes.byref(d), x.ptr, wdev.data_ptr(), bdev.data_ptr(),
                    res.ptr if res else None, out.ptr, self.sh),
                   flops=2 * n * ho * wo * cout * kh * kw * cin, nbytes=nbytes, keep=d)

    def _l_depthwise(self, ln):
        a = ln.attrs
        x = self._val(ln.inputs[0])
        if not x.dense:
            raise NotImplementedError('depthwise input must be dense')
        n, h, w, c = x.shape
        _, ho, wo, _ = ln.shape
        scale, bias = self._fold_bn(a, c)
        Wt = self.weights[a['weights']].astype(np.float64)[:, :, :, 0] * scale   # [kh,kw,c]
        wdev = self._dev_f32(Wt.reshape(a['kh'] * a['kw'], c))
        bdev = self._dev_f32(bias)
        self.param_bytes += a['kh'] * a['kw'] * c * 4
        out = Val(self._alloc(ln.shape, x.dt), ln.shape, x.dt)
        self.vals[ln] = out
        self._emit('depthwise', a['weights'].rsplit('/', 1)[0], 'fss_dwconv2d_nhwc',
                   (x.ptr, wdev.data_ptr(), bdev.data_ptr(), out.ptr, n, h, w, c, a['kh'], a['kw'],
                    a['stride'], a['rate'], a['pad_top'], a['pad_left'], ho, wo,
                    _lib.ACT_CODES[a['activation']], _CODE[x.dt], self.sh),
                   flops=2 * n * ho * wo * c * a['kh'] * a['kw'],
                   nbytes=(n * h * w * c + n * ho * wo * c) * _ESZ[x.dt])

    # ---- pooling
    def _l_maxpool(self, ln):
        a = ln.attrs
        x = self._val(ln.inputs[0])
        if not x.dense:
            raise NotImplementedError('pool input must be dense')
        n, h, w, c = x.shape
        _, ho, wo, _ = ln.shape
        out = Val(self._alloc(ln.shape, x.dt), ln.shape, x.dt)
        self.vals[ln] = out
        if a['k'] == 1:
            self._emit('subsample', 'subsample', 'fss_subsample_nhwc',
                       (x.ptr, out.ptr, n, h, w, c, a['stride'], ho, wo, _CODE[x.dt], self.sh),
                       nbytes=2 * n * ho * wo * c * _ESZ[x.dt])
        else:
            self._emit('maxpool', 'max_pool', 'fss_maxpool_nhwc',
                       (x.ptr, out.ptr, n, h, w, c, a['k'], a['stride'], a['pad_top'],
                        a['pad_left'], ho, wo, _CODE[x.dt], self.sh),
                       nbytes=(n * h * w * c + n * ho * wo * c) * _ESZ[x.dt])

    def _l_avgpool(self, ln):
        if ln in self.vals:                 # already produced with its siblings (below)
            return
        a = ln.attrs
        x = self._val(ln.inputs[0])
        if not x.dense:
            raise NotImplementedError('pool input must be dense')
        n, h, w, c = x.shape
        # the levels of a pyramid pooling module read the same tensor: one pass for all
        sibs = [s for s in ln.inputs[0].consumers if s.kind == 'avg_pool' and s not in self.vals]
        if 2 <= len(sibs) <= 4 and self.fuse_pyramid_pool:
            L = len(sibs)
            outs = []
            for s_ in sibs:
                v = Val(self._alloc(s_.shape, x.dt), s_.shape, x.dt)
                self.vals[s_] = v
                outs.append(v)
            ptrs = (ctypes.c_void_p * L)(*[v.ptr for v in outs])
            kh = (ctypes.c_int * L)(*[s_.attrs['kh'] for s_ in sibs])
            kw = (ctypes.c_int * L)(*[s_.attrs['kw'] for s_ in sibs])
            bh = (ctypes.c_int * L)(*[s_.shape[1] for s_ in sibs])
            bw = (ctypes.c_int * L)(*[s_.shape[2] for s_ in sibs])
            cols = sum(s_.shape[2] for s_ in sibs)
            work = torch.empty((n * h * cols * c,), dtype=torch.float32, device=self.device)
            self.workspaces.append(work)
            self._emit('avgpool', 'avg_pool_pyramid', 'fss_avgpool_pyramid_nhwc',
                       (x.ptr, ptrs, kh, kw, bh, bw, L, work.data_ptr(), work.numel() * 4, n, h, w, c,
                        _CODE[x.dt], self.sh),
                       nbytes=n * h * w * c * _ESZ[x.dt] + 2 * work.numel() * 4,
                       keep=(ptrs, kh, kw, bh, bw))
            return
        _, bh, bw, _ = ln.shape
        out = Val(self._alloc(ln.shape, x.dt), ln.shape, x.dt)
        self.vals[ln] = out
        self._emit('avgpool', 'avg_pool', 'fss_avgpool_bins_nhwc',
                   (x.ptr, out.ptr, n, h, w, c, a['kh'], a['kw'], bh, bw, _CODE[x.dt], self.sh),
                   nbytes=(n * bh * a['kh'] * bw * a['kw'] * c + n * bh * bw * c) * _ESZ[x.dt])

    # ---- resize (deferred: fused into its consumer when possible)
    def _l_resize(self, ln):
        src = ln.inputs[0]
        align = ln.attrs['align_corners']

        def materialise():
            x = self._val(src)
            if not x.dense:
                raise NotImplementedError('resize input must be dense')
            n, h, w, c = x.shape
            _, ho, wo, _ = ln.shape
            place = self.placement.get(ln)
            if place is not None and x.phys_c == c:
                buf, off, ld = place
                out = Val(buf, ln.shape, x.dt, phys_c=c, ld=ld, coff=off)
            else:
                self.placement.pop(ln, None)
                out = Val(self._alloc(ln.shape, x.dt, phys_c=x.phys_c), ln.shape, x.dt,
                          phys_c=x.phys_c)
            self.vals[ln] = out
            self._emit('resize', 'resize_bilinear', 'fss_resize_bilinear_nhwc',
                       (x.ptr, out.ptr, n, h, w, x.phys_c, ho, wo, out.ld, 1 if align else 0,
                        _CODE[x.dt], self.sh),
                       nbytes=(n * h * w + n * ho * wo) * x.phys_c * _ESZ[x.dt])

        self.lazy[ln] = ('resize', src, align, materialise)

    # ---- add / relu
    def _pyramid_term(self, ln, like):
        """ln is a deferred align_corners resize of a small dense map shaped like `like`?"""
        d = self.lazy.get(ln)
        if not d or d[0] != 'resize' or not d[2] or ln.is_output or len(ln.consumers) != 1:
            return None
        small = self._val(d[1])
        if (small.dense and small.dt == self.dt and small.shape[3] % _EPC[small.dt] == 0
                and small.shape[1] * small.shape[2] <= 256 and ln.shape == like.shape):
            return small
        return None

    def _l_add(self, ln):
        a_n, b_n = ln.inputs
        term = self._pyramid_term(b_n, a_n)
        if term is None:
            term = self._pyramid_term(a_n, b_n)
            if term is not None:
                a_n, b_n = b_n, a_n
        if term is not None:
            self.lazy.pop(b_n)
            prev = self.lazy.get(a_n)
            if prev and prev[0] == 'pyramid' and not a_n.is_output and len(a_n.consumers) == 1:
                self.lazy.pop(a_n)
                base, levels = prev[1], prev[2] + [term]
            else:
                base, levels = a_n, [term]
            if len(levels) <= 8:
                def materialise(base=base, levels=levels):
                    x = self._val(base)
                    if not x.dense:
                        raise NotImplementedError('pyramid base must be dense')
                    out = Val(self._alloc(ln.shape, x.dt), ln.shape, x.dt)
                    self.vals[ln] = out
                    n, h, w, c = x.shape
                    L = len(levels)
                    ptrs = (ctypes.c_void_p * L)(*[t.ptr for t in levels])
                    bh = (ctypes.c_int * L)(*[t.shape[1] for t in levels])
                    bw = (ctypes.c_int * L)(*[t.shape[2] for t in levels])
                    self._emit('pyramid', 'pyramid_add', 'fss_pyramid_add_nhwc',
                               (x.ptr, ptrs, bh, bw, L, out.ptr, n, h, w, c, _CODE[x.dt], self.sh),
                               nbytes=2 * n * h * w * c * _ESZ[x.dt], keep=(ptrs, bh, bw))
                self.lazy[ln] = ('pyramid', base, levels, materialise)
                return
        a, b = self._val(ln.inputs[0]), self._val(ln.inputs[1])
        if not (a.dense and b.dense) or a.dt != b.dt or a.phys_c != a.shape[3]:
            raise NotImplementedError('add needs dense operands of one dtype')
        act, result = None, ln
        relu = self._only_consumer(ln, 'relu')
        if relu is not None:
            act, result = 'relu', relu
            self.fused.add(relu)
        out = Val(self._alloc(ln.shape, a.dt), ln.shape, a.dt)
        self.vals[result] = out
        self.vals[ln] = out
        count = int(np.prod(a.shape))
        self._emit('add', 'add', 'fss_add_act',
                   (a.ptr, b.ptr, out.ptr, count, _lib.ACT_CODES[act], _CODE[a.dt], self.sh),
                   nbytes=3 * count * _ESZ[a.dt])

    def _l_relu(self, ln):
        raise NotImplementedError('stand-alone relu (not following an add) is not supported')

    # ---- concat
    def _l_concat(self, ln):
        out = self.vals[ln]
        n, h, w, ctot = out.shape
        off = 0
        for src in ln.inputs:
            c = src.shape[3]
            x = self._val(src)      # placed producers materialise straight into `out`
            if not (x.buf is out.buf and x.coff == off):
                if x.dt != out.dt or x.phys_c != c:
                    raise NotImplementedError('concat inputs must share the compute dtype')
                esz = _ESZ[x.dt]
                self._emit('copy', 'concat_copy', 'fss_copy_channels',
                           (x.ptr, x.ld, out.buf.data_ptr() + off * esz, ctot, n * h * w, c,
                            _CODE[x.dt], self.sh), nbytes=2 * n * h * w * c * esz)
            off += c

    # ---- argmax / lut / crop
    def _l_argmax(self, ln):
        src = ln.inputs[0]
        d = self.lazy.get(src)
        full_ptr = None
        n, ho, wo, _ = ln.shape
        if d and d[0] == 'resize' and d[2]:
            # Two align_corners zooms in a row, the first by a whole factor (its output keeps
            # every source pixel as a knot): inside one source cell the first zoom samples a
            # bilinear function, which the second reproduces exactly from those samples, so
            # the pair equals one zoom from the source (to float rounding).
            inner = d[1]
            d0 = self.lazy.get(inner)
            if (self.collapse_zooms and d0 and d0[0] == 'resize' and d0[2] and not inner.is_output
                    and len(inner.consumers) == 1 and not src.is_output and len(src.consumers) == 1):
                (h0, w0), (h1, w1) = d0[1].shape[1:3], inner.shape[1:3]
                if h0 > 1 and w0 > 1 and (h1 - 1) % (h0 - 1) == 0 and (w1 - 1) % (w0 - 1) == 0:
                    self.lazy.pop(inner)
                    d = (d[0], d0[1]) + tuple(d[2:])
        if d and d[0] == 'resize' and d[2] and self._val(d[1]).dt == 'f32':
            x = self._val(d[1])
            if src.is_output or len(src.consumers) > 1:
                # the zoomed logits are wanted too: the fused kernel stores them
                self.lazy.pop(src)
                full = Val(self._alloc(src.shape, 'f32'), src.shape, 'f32')
                self.vals[src] = full
                full_ptr = full.ptr
        else:
            x = self._val(src)
        if x.dt != 'f32' or not x.dense:
            raise NotImplementedError('argmax expects dense float32 logits')
        _, h, w, c = x.shape
        labels = Val(torch.empty((n, ho, wo, 1), dtype=torch.uint8, device=self.device),
                     (n, ho, wo, 1), 'u8')
        self.vals[ln] = labels
        self._emit('argmax', 'upsample_argmax', 'fss_upsample_argmax',
                   (x.ptr, labels.ptr, full_ptr, n, h, w, c, x.ld, ho, wo, self.sh),
                   nbytes=n * h * w * c * 4 + n * ho * wo + (n * ho * wo * c * 4 if full_ptr else 0))

    def _l_lut(self, ln):
        labels = self._val(ln.inputs[0])
        if labels.dt != 'u8':
            raise NotImplementedError('lut expects a label map')
        n, h, w, _ = labels.shape
        table = np.asarray(ln.attrs['table'], np.float32)
        tdev = self._dev_f32(table)
        ho, wo, result = h, w, ln
        crop = self._only_consumer(ln, 'crop')
        if crop is not None:
            _, ho, wo, _ = crop.shape
            result = crop
            self.fused.add(crop)
        ch = table.shape[1]
        out = Val(self._alloc((n, ho, wo, ch), 'f32'), (n, ho, wo, ch), 'f32')
        self.vals[result] = out
        self._emit('lut', 'label_lut', 'fss_label_lut',
                   (labels.ptr, out.ptr, tdev.data_ptr(), n, h, w, ho, wo, ch, table.shape[0],
                    self.sh), nbytes=n * ho * wo * (1 + 4 * ch))

    def _l_crop(self, ln):
        # not fused into a lut: a top-left view, resolved when the value is fetched
        src = self._val(ln.inputs[0])
        self.vals[ln] = Val(src.buf, ln.shape, src.dt, phys_c=src.phys_c, ld=src.ld, coff=src.coff)

    # ------------------------------------------------------------ conv tuning
    _TUNE_CACHE = None      # {key string: config}, shared by every plan of the process

    @staticmethod
    def _tune_file():
        return os.environ.get('FSS_TUNE_FILE',
                              os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_tuned.json'))

    @classmethod
    def _tune_stamp(cls):
        # library version + the whole configuration table: a table measured with other
        # kernels or other tile shapes is not reused
        info, parts = (ctypes.c_int * 7)(), [str(lib.fss_version())]
        for cfg in range(1, lib.fss_conv_config_count() + 1):
            _lib.call('fss_conv_config_info', cfg, info)
            parts.append('x'.join(str(v) for v in info))
        return '/'.join(parts)

    @classmethod
    def _tune_load(cls):
        """Pre-tuned choices shipped next to this module (ignored when they were
        measured with another kernel table)."""
        cls._TUNE_CACHE = {}
        try:
            with open(cls._tune_file()) as f:
                d = json.load(f)
            if d.get('stamp') == cls._tune_stamp():
                cls._TUNE_CACHE.update({k: int(v) for k, v in d['entries'].items()})
        except (OSError, ValueError, KeyError):
            pass

    @classmethod
    def save_tuning(cls, path=None):
        """Writes every choice made so far (tools/tune_convs.py calls this)."""
        with open(path or cls._tune_file(), 'w') as f:
            json.dump({'stamp': cls._tune_stamp(), 'entries': dict(sorted((cls._TUNE_CACHE or {}).items()))},
                      f, indent=0)

    def _autotune_convs(self, warm=2, reps=int(os.environ.get('FSS_TUNE_REPS', '4'))):
        """Picks the fastest tile configuration of the conv kernel for each distinct
        layer shape: from the pre-tuned table when the shape is in it, else by timing
        the candidates (HIP events on the plan's stream).  Only configurations of one
        MFMA shape (self.mfma_edge) are candidates: inside a shape class results are
        bit-identical, so the choice affects speed and never the output."""
        if CompiledGraph._TUNE_CACHE is None:
            self._tune_load()
        cache = CompiledGraph._TUNE_CACHE
        ncfg = lib.fss_conv_config_count()
        info = (ctypes.c_int * 7)()
        # candidates of the plan's MFMA shape; the split-bf16 launches take only those with an
        # even number of k steps per tile
        all_cands, split_cands, wide_cands, tile_n = [], [], [], {}
        small_all, small_split = [], []
        for cfg in range(1, ncfg + 1):
            _lib.call('fss_conv_config_info', cfg, info)
            if info[6] != self.mfma_edge:
                continue
            even = (info[4] // (16 * (64 // info[6]))) % 2 == 0
            all_cands.append(cfg)
            tile_n[cfg] = info[1]
            # two weight tiles per stage (FSS_F16_W2): the ring must still fit the LDS
            if info[5] * (info[0] + 2 * info[1]) * info[4] < 160 * 1024 and cfg != 35:
                wide_cands.append(cfg)
            if even:
                split_cands.append(cfg)
            # small-tile configurations: the ones a split-K launch is tried with
            if info[0] * info[1] <= 128 * 64:
                small_all.append(cfg)
                if even:
                    small_split.append(cfg)
        e0 = e1 = None
        ms = ctypes.c_float()
        tune_ws = None
        for idx, (s, fname) in enumerate(zip(self.steps, self.step_fname)):
            if s.kind != 'conv':
                continue
            d = s.keep[0] if isinstance(s.keep, tuple) else s.keep
            key = ','.join(str(getattr(d, f[0])) for f in d._fields_ if f[0] != 'config')
            x3 = d.in_dtype == _lib.F32_BF16X3
            candidates, small = (split_cands, small_split) if x3 else (all_cands, small_all)
            if d.in_dtype == _lib.F16_W2:
                candidates = wide_cands
            cands = candidates
            # split-K candidates: an output too small to fill the chip (at most four 64x64 tiles
            # per CU) under a long K axis, i.e. the late backbone blocks at one or two frames
            m_rows = d.n * d.ho * d.wo
            ksplits = ()
            if (fname == 'fss_conv2d_nhwc' and self.ksplit and m_rows * d.cout <= 64 * 64 * 256 * 4
                    and d.kh * d.kw * d.cin >= 1024):
                ksplits = (2, 4, 8)
            if fname == 'fss_conv2d_dual_nhwc':
                key += ',%d,%d,dual%d' % (s.args[6] is not None, self.mfma_edge, s.args[3])
            elif fname == 'fss_conv2d_dual_head_nhwc':
                # the N tile must hold every output channel (16x16 MFMA shape only)
                key += ',head%d,%d,dual%d,%d' % (s.args[12], self.mfma_edge, s.args[3],
                                                 1 if s.args[6] is not None else 2 if s.args[7] is not None else 0)
                cands = [c for c in candidates if tile_n[c] >= max(d.cout, 128)]
            else:
                key += ',%d,%d' % (s.args[4] is not None, self.mfma_edge)
            if ksplits:
                key += ',ksplit'           # timed with the split-K candidates as well
            best = cache.get(key)
            if best is None:
                if e0 is None:
                    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                    _lib.call('fss_event_create', ctypes.byref(e0))
                    _lib.call('fss_event_create', ctypes.byref(e1))
                best, best_ms = 0, float('inf')
                trials = [(cfg, 1) for cfg in cands] + [(cfg, k) for k in ksplits for cfg in small]
                if ksplits:
                    need = max(ksplits) * m_rows * ((d.cout + 3) // 4 * 4) * 4
                    if tune_ws is None or tune_ws.numel() * 4 < need:
                        tune_ws = torch.empty(need // 4, dtype=torch.float32, device=self.device)
                for cfg, k in trials:
                    d.config = cfg
                    fn, args = s.fn, s.args
                    if k > 1:
                        fn = lib.fss_conv2d_splitk_nhwc
                        args = s.args[:6] + (k, tune_ws.data_ptr(), tune_ws.numel() * 4, s.args[6])
                    for _ in range(warm):
                        check(fn(*args), s.name)
                    _lib.call('fss_event_record', e0, self.sh)
                    for _ in range(reps):
                        check(fn(*args), s.name)
                    _lib.call('fss_event_record', e1, self.sh)
                    _lib.call('fss_event_sync', e1)
                    _lib.call('fss_event_elapsed_ms', e0, e1, ctypes.byref(ms))
                    if ms.value < best_ms:
                        best, best_ms = cfg + 1000 * (k if k > 1 else 0), ms.value
                cache[key] = best
                self.tuned_now += 1
            d.config = best % 1000
            if best >= 1000 and fname == 'fss_conv2d_nhwc':
                # the split-K form of this launch, with a workspace of its own (the slabs are
                # written and read by this step only)
                k = best // 1000
                ws = torch.empty(k * m_rows * ((d.cout + 3) // 4 * 4), dtype=torch.float32, device=self.device)
                self.workspaces.append(ws)
                s.fn = lib.fss_conv2d_splitk_nhwc
                s.args = s.args[:6] + (k, ws.data_ptr(), ws.numel() * 4, s.args[6])
                self.step_fname[idx] = 'fss_conv2d_splitk_nhwc'
        if e0 is not None:
            _lib.call('fss_event_destroy', e0)
            _lib.call('fss_event_destroy', e1)

    # ------------------------------------------------------------ two streams
    def _dependencies(self):
        """deps[i] = steps that must finish before step i starts, found from the buffers each
        launch reads and writes.  Pointer arguments are recognised by the entry point's
        prototype; _OUT_ARGS says which of them are written.  Every pointer must fall inside
        a buffer the plan knows - an activation (vals / temps / input), a workspace, or a
        constant uploaded at compile time (weights, biases, tables: read-only, no edges) -
        otherwise the schedule could miss an edge, and the plan build fails."""
        def spans(tensors):
            out, seen = [], set()
            for t in tensors:
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    out.append((t.data_ptr(), t.data_ptr() + max(t.numel() * t.element_size(), 1)))
            out.sort()
            return out, [r[0] for r in out]

        # (shared blocks first: a block is one buffer whoever its tenants are)
        ranges, starts = spans([b[0] for b in self._blocks]
                               + [v.buf for v in list(self.vals.values()) + self.temps + [self.input_val]]
                               + self.workspaces)
        consts, cstarts = spans(self.keepalive)

        def find(table, first, ptr):
            k = bisect.bisect_right(first, ptr) - 1
            return k if k >= 0 and ptr < table[k][1] else None

        last_writer, readers, deps, main_src = {}, {}, [], []
        for i, (s, fname) in enumerate(zip(self.steps, self.step_fname)):
            outs = _OUT_ARGS[fname]
            argtypes = _lib.PROTOTYPES[fname][1]
            rd, wr = [], []
            for ai, arg in enumerate(s.args[:-1]):
                if isinstance(arg, ctypes.Array) and getattr(arg, '_type_', None) is ctypes.c_void_p:
                    ptrs = [q for q in arg if q]
                elif argtypes[ai] is ctypes.c_void_p and isinstance(arg, int) and arg:
                    ptrs = [arg]
                else:
                    continue
                for q in ptrs:
                    b = find(ranges, starts, q)
                    if b is not None:
                        (wr if ai in outs else rd).append(b)
                    elif find(consts, cstarts, q) is None or ai in outs:
                        raise RuntimeError('%s (%s): pointer argument %d is outside every buffer of the plan'
                                           % (fname, s.name, ai))
            d = set()
            for b in rd + wr:
                if b in last_writer:
                    d.add(last_writer[b])
            for b in wr:                      # write after read (slices of a concat buffer)
                d |= readers.get(b, set())
            d.discard(i)
            deps.append(d)
            main_src.append(last_writer.get(rd[0]) if rd else None)
            for b in rd:
                readers.setdefault(b, set()).add(i)
            for b in wr:
                last_writer[b] = i
                readers[b] = set()
        return deps, main_src

    def _assign_streams(self, release_after=None):
        """Puts the second chain that starts from the input frame (ICNet's full-resolution
        branch) on a side stream, with events where the chains exchange data.
        release_after (tests and experiments): index of a step of the main chain the side
        chain waits for before it starts, i.e. a fork in the middle of the pass."""
        deps, main_src = self._dependencies()
        for s in self.steps:
            s.stream, s.waits, s.signal = 0, [], None
        roots = 0
        for i, s in enumerate(self.steps):
            if not deps[i]:
                s.stream = 0 if roots == 0 else 1
                roots += 1
            else:
                src = main_src[i] if main_src[i] is not None else max(deps[i])
                s.stream = self.steps[src].stream
        side = [i for i, s in enumerate(self.steps) if s.stream == 1]
        if not side:
            return False
        if release_after is not None:
            deps[side[0]] = set(deps[side[0]]) | {release_after}
        synced = {}                               # (from stream, to stream) -> newest step waited for

        def event_of(j):
            if self.steps[j].signal is None:
                e = ctypes.c_void_p()
                _lib.call('fss_event_create', ctypes.byref(e))
                self.events.append(e)
                self.steps[j].signal = e
            return self.steps[j].signal

        for i, s in enumerate(self.steps):
            for j in sorted(deps[i], reverse=True):
                sj = self.steps[j].stream
                if sj != s.stream and synced.get((sj, s.stream), -1) < j:
                    s.waits.append(event_of(j))
                    synced[(sj, s.stream)] = j
        self.join_after = side[-1]
        event_of(self.join_after)
        return True

    # -------------------------------------------------------------- execution
    def run_eager(self):
        """One forward pass, launch by launch.  With a side stream: it forks from the
        plan's stream at the start and joins it at the end, so for callers the whole pass
        is still ordered on the plan's stream."""
        if not self.two_streams:
            for s in self.steps:
                check(s.fn(*s.args), s.kind + ':' + s.name)
            return
        handles = (self.sh, self.sh_side)
        _lib.call('fss_event_record', self.fork_event, self.sh)
        _lib.call('fss_stream_wait_event', self.sh_side, self.fork_event)
        for s in self.steps:
            h = handles[s.stream]
            for e in s.waits:
                _lib.call('fss_stream_wait_event', h, e)
            check(s.fn(*(s.args[:-1] + (h,))), s.kind + ':' + s.name)
            if s.signal is not None:
                _lib.call('fss_event_record', s.signal, h)
        _lib.call('fss_stream_wait_event', self.sh, self.steps[self.join_after].signal)

    def _capture(self):
        if self.graph_exec is not None:
            # an executable graph may be torn down only when nothing of its last launch is
            # still in flight on EITHER stream: the plan's stream alone being idle says
            # nothing about the runtime's bookkeeping for the side branch, so the whole
            # device is drained first
            _lib.call('fss_device_sync')
            _lib.call('fss_graph_destroy', self.graph_exec)
            self.graph_exec = None
        _lib.call('fss_capture_begin', self.sh)
        ex = ctypes.c_void_p()
        try:
            self.run_eager()
        finally:
            _lib.call('fss_capture_end', self.sh, ctypes.byref(ex))
        self.graph_exec = ex

    def _time_launches(self, reps=5):
        for _ in range(2):
            self.launch()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.launch()
        self.sync()
        return (time.perf_counter() - t0) / reps

    def _setup_streams(self, tune=True):
        """Side-stream schedule: the side chain forks from the plan's stream at the start of
        the pass and joins at the first launch that needs its result.  The two-stream graph
        is kept if it is faster than the single-stream one; its replays must reproduce the
        eager single-stream outputs bit for bit, and a mismatch is an error (it would mean a
        missing dependency edge or a kernel that misbehaves beside another one), never a
        silent fallback.  `stream_choice` records what was decided and why."""
        self.side_stream = torch.cuda.Stream(device=self.device)
        self.sh_side = ctypes.c_void_p(self.side_stream.cuda_stream)
        self.fork_event = ctypes.c_void_p()
        _lib.call('fss_event_create', ctypes.byref(self.fork_event))
        self.events.append(self.fork_event)
        if not self._assign_streams(None):
            self.stream_choice = 'single: the graph has no second chain'
            return False
        if not tune:
            self.two_streams = True
            self.stream_choice = 'two: not timed'
            return True
        # reference: the eager single-stream pass that has just run (same input buffer)
        self.sync()
        outs = [self.vals[ln] for ln in self.out_nodes.values()]
        ref = [v.buf.clone() for v in outs]
        self.two_streams = False
        self._capture()
        single_t = self._time_launches()
        self.two_streams = True
        self._capture()
        two_t = self._time_launches()
        for _ in range(3):
            self.launch()
            self.sync()
            for v, r in zip(outs, ref):
                if not torch.equal(v.buf.view(torch.uint8), r.view(torch.uint8)):
                    raise RuntimeError('replays of the two-stream graph differ from the eager '
                                       'single-stream pass')
        if os.environ.get('FSS_PLAN_DEBUG'):
            print('two streams: %.3f ms, single stream %.3f ms' % (two_t * 1e3, single_t * 1e3), flush=True)
        self.two_streams = two_t < single_t
        self.stream_choice = '%s: two streams %.3f ms, one stream %.3f ms' % (
            'two' if self.two_streams else 'single', two_t * 1e3, single_t * 1e3)
        self.release_after = None if self.two_streams else 'single'
        return self.two_streams

    def launch(self):
        """Enqueues one forward pass on the plan's stream (asynchronous)."""
        if self.graph_exec is not None:
            _lib.call('fss_graph_launch', self.graph_exec, self.sh)
        else:
            self.run_eager()

    def sync(self):
        self.stream.synchronize()

    def set_input(self, images_u8):
        """Copies a uint8 [n,h,w,3] (or [h,w,3]) host array into the input buffer."""
        a = np.ascontiguousarray(images_u8, dtype=np.uint8)
        if a.ndim == 3:
            a = a[None]
        if tuple(a.shape) != tuple(self.input_val.shape):
            raise ValueError('input shape %s does not match the placeholder %s'
                             % (a.shape, self.input_val.shape))
        with torch.cuda.stream(self.stream):
            self.input_val.buf.copy_(torch.from_numpy(a))

    def value(self, tensor):
        """Device value (`Val`) of a graph tensor given in `outputs`."""
        return self.vals[self.out_nodes[tensor.node.index]]

    def fetch(self, tensor):
        """Device -> host copy of an output tensor as a numpy array (logical shape)."""
        v = self.value(tensor)
        self.sync()
        n, h, w, c = v.shape
        t = v.buf.reshape(v.buf.shape[0], v.buf.shape[1], v.buf.shape[2], -1)
        with torch.cuda.stream(self.stream):
            t = t[:, :h, :w, v.coff:v.coff + c]
            if t.dtype in (torch.bfloat16, torch.float16):
                t = t.float()
            if v.dt == 'f32s':
                # whole groups of 8 channels, decoded on the host
                c0, c1 = v.coff // 8 * 8, -(-(v.coff + c) // 8) * 8
                raw = v.buf.reshape(v.buf.shape[0], v.buf.shape[1], v.buf.shape[2], -1)[:, :h, :w, c0:c1].cpu().numpy()
                return np.ascontiguousarray(split_bf16_decode(raw)[..., v.coff - c0:v.coff - c0 + c])
            out = t.cpu()
        return out.numpy()

    def run(self, images_u8, fetch=None):
        """set_input + launch + fetch (all outputs by default)."""
        self.set_input(images_u8)
        self.launch()
        return [self.fetch(t) for t in (fetch if fetch is not None else self.outputs)]

    def __del__(self):
        try:
            lib.fss_device_sync()
            if self.graph_exec is not None:
                lib.fss_graph_destroy(self.graph_exec)
            for e in self.events:
                lib.fss_event_destroy(e)
        except Exception:
            pass

    # -------------------------------------------------------------- profiling
    def profile(self, repeats=3):
        """Times every launch with HIP events recorded on the plan's stream
        (eager mode, single stream).  Returns a list of dicts (kind, name, ms, mean_ms, flops,
        bytes): ms = the best, mean_ms = the mean of `repeats` passes."""
        n = len(self.steps)
        evs = []
        for _ in range(n + 1):
            e = ctypes.c_void_p()
            _lib.call('fss_event_create', ctypes.byref(e))
            evs.append(e)
        best = [float('inf')] * n
        total = [0.0] * n
        for _ in range(repeats):
            _lib.call('fss_event_record', evs[0], self.sh)
            for k, s in enumerate(self.steps):
                check(s.fn(*s.args), s.name)
                _lib.call('fss_event_record', evs[k + 1], self.sh)
            _lib.call('fss_event_sync', evs[n])
            for k in range(n):
                ms = ctypes.c_float()
                _lib.call('fss_event_elapsed_ms', evs[k], evs[k + 1], ctypes.byref(ms))
                best[k] = min(best[k], ms.value)
                total[k] += ms.value
        for e in evs:
            _lib.call('fss_event_destroy', e)
        return [dict(kind=s.kind, name=s.name, ms=best[k], mean_ms=total[k] / repeats, flops=s.flops, graph_flops=s.graph_flops, bytes=s.bytes,
                     mfma_per_product=s.mpp, fname=self.step_fname[k],
                     config=((s.keep[0] if isinstance(s.keep, tuple) else s.keep).config
                             + (1000 * s.args[6] if self.step_fname[k] == 'fss_conv2d_splitk_nhwc' else 0)
                             if s.kind == 'conv' else None))
                for k, s in enumerate(self.steps)]
