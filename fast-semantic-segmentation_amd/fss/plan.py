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

The compiler is spread over five modules: this one (the class, graph import and rewrites, the
lowering of everything but convolutions, the execution API and profiling), fss.plan_conv (the
convolutions), fss.plan_tune (tile configurations), fss.plan_schedule (buffers, dependencies,
streams, capture) and fss.plan_base / fss.packing (shared tables, classes and bit formats).
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

from fss.packing import (f32_to_bf16_bits, f32_to_f16_bits, f32_split_bf16_rows, f32_split_f16_rows,
                         split_bf16_decode, split_bf16_encode, stem_pack_f16x2)
from fss.plan_base import (_CODE, _EPC, _ESZ, _OUT_ARGS, _TORCH_DT, DEFAULT_HALF_ACT_SCOPES,
                           DEFAULT_HALF_SCOPES, LNode, Step, Val)
from fss.plan_conv import ConvLowering
from fss.plan_schedule import Schedule
from fss.plan_tune import ConvTuning


class CompiledGraph(ConvLowering, ConvTuning, Schedule):
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
        if a['biases'] is not None or ln.inputs[0].shape[3] % 8 or ln.shape[3] % 8:
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
        if (ln.inputs[0].shape[3] % 64 or (a['kh'], a['kw'], a['stride'], a['rate']) != (1, 1, 1, 1)
                or ln.shape[:3] != ln.inputs[0].shape[:3]):
            return False    # a shared launch: 1x1, whole K tiles of both inputs
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
            if c.kind == 'depthwise' and not c.is_output and self._store_dt(c, in_dt) == 'f16':
                continue        # float arithmetic on either side: only the two stores round
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
        # (a half-precision input means that every reader of the output takes half precision too)
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
