"""Lowering of the convolutions (fss.plan.CompiledGraph): batch-norm folding, the uint8 stem
forms, two-input / chained / classifier-head launches, convolutions moved in front of a zoom or
of the pyramid sum, and the plain implicit-GEMM launch with its precision modes."""
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


class ConvLowering(object):
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
            # whole K tiles of both inputs - unless `ln` runs in front of its zoom, in launches of
            # its own, and the 1x1 half only takes its zoomed result as a residual
            if self._zoom_source(ln) is None or ln.inputs[0].shape[3] % 8 or A.inputs[0].shape[3] % 8:
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
        if (x.dt == 'f32s' and x.dense and x.phys_c == cin and kh * kw * cin >= 8192
                and self._takes_half_acts(ln)):
            # a long K axis repays a pass that rounds the input to half precision: two MFMAs per
            # product (two-f16 weights) instead of three
            xh = Val(self._alloc(x.shape, 'f16'), x.shape, 'f16')
            self.temps.append(xh)
            self._emit('copy', a['weights'].rsplit('/', 1)[0] + '/to_half', 'fss_to_half',
                       (x.ptr, xh.ptr, n * h * w * cin, _CODE[x.dt], self.sh), nbytes=n * h * w * cin * 6)
            x = xh
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
        if (self.resident_3x3 and (kh, kw, a['stride'], a['rate'], a['pad_top'], a['pad_left']) == (3, 3, 1, 1, 1, 1)
                and pc == cin == 128 and cout == 256 and (ho, wo) == (h, w) and in_dt in ('bf16', 'f16')
                and out_dt == in_dt and res is None and a['biases'] is None and k_pad == 9 * 128
                and x.ld % 8 == 0 and h * w * x.ld * 2 < 2 ** 31 - 1):
            # 128 -> 256 channels at stride 1 (`Conv_3`): persistent blocks, the input tile and its
            # halo in LDS for all nine taps, the filter streamed as ready-made LDS images (measured
            # 3.33 against 3.93 ms for the best implicit-GEMM tile at 192 frames; same bits)
            idx = _lib.conv3x3_c128_index().astype(np.int64)
            flat = torch.from_numpy(idx[:, 0] * k_pad + idx[:, 1]).to(self.device)
            wimg = wdev.flatten()[flat].contiguous()
            self.keepalive.append(wimg)
            out = Val(self._alloc(ln.shape, out_dt), ln.shape, out_dt)
            self.vals[result] = out
            self._emit('conv3x3', a['weights'].rsplit('/', 1)[0], 'fss_conv3x3_c128_nhwc',
                       (x.ptr, x.ld, wimg.data_ptr(), bdev.data_ptr(), out.ptr, out.ld, n, h, w,
                        _lib.ACT_CODES[act], _CODE[in_dt], self.sh),
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
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=kh, kw=kw, stride=a['stride'],
                              dilation=a['rate'], pad_top=a['pad_top'], pad_left=a['pad_left'], ho=ho, wo=wo,
                              act=_lib.ACT_CODES[act], in_dtype=_lib.F16_W2,
                              out_dtype=_CODE[out_dt], y_ld=out.ld, res_ld=res.ld if res else 0,
                              k_pad=k_pad, cout_pad=cout_pad)
            nbytes = ((n * h * w * cin + 2 * cout * kh * kw * cin) * 2 + n * ho * wo * cout * _ESZ[out_dt]
                      + (n * ho * wo * cout * _ESZ[res.dt] if res else 0))
            self._emit('conv', a['weights'].rsplit('/', 1)[0], 'fss_conv2d_nhwc',
                       (ctypes.byref(d), x.ptr, self._pack(Wp, 'f16w').data_ptr(), bdev.data_ptr(),
                        res.ptr if res else None, out.ptr, self.sh),
                       flops=2 * n * ho * wo * cout * kh * kw * cin, nbytes=nbytes, keep=d, mpp=2)
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
