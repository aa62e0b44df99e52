"""Choice of the tile configuration of every convolution launch (fss.plan.CompiledGraph): the
shipped table (conv_tuned.json), its stamp, and timing of the candidates for shapes not in it."""
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


class ConvTuning(object):
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
        if type(self)._TUNE_CACHE is None:
            self._tune_load()
        cache = type(self)._TUNE_CACHE
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
            if (fname == 'fss_conv2d_nhwc' and self.ksplit and d.in_dtype != _lib.F16_W2
                    and m_rows * d.cout <= 64 * 64 * 256 * 4
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
