"""Device memory and launch order of a plan (fss.plan.CompiledGraph): activation buffers and
their sharing, the dependencies between launches, the two-stream schedule, eager execution and
hipGraph capture."""
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


class Schedule(object):
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
