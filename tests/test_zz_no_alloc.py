"""fss_avgpool_pyramid_nhwc_static (row sums in the library's static device array, no
workspace argument, allocates nothing) against fss_avgpool_pyramid_nhwc (caller workspace):
bit-identical bin maps on two calls in a row, the second with a smaller input, and a clear
error beyond FSS_PYRAMID_STATIC_MAX_BYTES.  Both calls go through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pool_both(L, xd, levels, code, tdt):
    """Runs the workspace form and the static form on the same device tensor; returns the two
    lists of bin maps as raw bytes."""
    n, h, wd, c = xd.shape
    Lv = len(levels)
    khs = (ctypes.c_int * Lv)(*[l[0] for l in levels])
    kws = (ctypes.c_int * Lv)(*[l[1] for l in levels])
    bhs = (ctypes.c_int * Lv)(*[h // l[0] for l in levels])
    bws = (ctypes.c_int * Lv)(*[wd // l[1] for l in levels])

    def outputs():
        outs = [torch.full((n, h // kh, wd // kw, c), -77.0, dtype=tdt, device='cuda') for kh, kw in levels]
        return outs, (ctypes.c_void_p * Lv)(*[o.data_ptr() for o in outs])

    ref, ref_ptrs = outputs()
    work = torch.empty((n * h * sum(bws) * c,), dtype=torch.float32, device='cuda')
    L.call('fss_avgpool_pyramid_nhwc', xd.data_ptr(), ref_ptrs, khs, kws, bhs, bws, Lv, work.data_ptr(),
           work.numel() * 4, n, h, wd, c, code, None)
    got, got_ptrs = outputs()
    torch.cuda.synchronize()
    L.call('fss_avgpool_pyramid_nhwc_static', xd.data_ptr(), got_ptrs, khs, kws, bhs, bws, Lv,
           n, h, wd, c, code, None)
    L.call('fss_device_sync')
    as_bytes = lambda ts: [t.contiguous().view(torch.uint8).cpu().numpy() for t in ts]
    return as_bytes(ref), as_bytes(got)


@pytest.mark.parametrize('dt', ['bf16', 'f32'])
def test_static_scratch_pyramid_pool_matches_workspace_form(dt):
    import gpu_util as gu
    from fss import _lib as L
    code, tdt, epc = gu.DT[dt]
    g = torch.Generator(device='cuda').manual_seed(7)
    # first call: large enough for the whole-row first stage (n*h*chunks >= 98304) and close to
    # the size limit; second call: a smaller input (column-piece first stage), which must not
    # see anything the first call left in the array
    c1 = 2048 if dt == 'bf16' else 1024
    for shape, factors in (((12, 33, 65, c1), (1, 2, 3, 4)), ((1, 18, 30, 512), (1, 2, 3, 6))):
        n, h, wd, c = shape
        levels = [(int(h / f), int(wd / f)) for f in factors]
        need = n * h * sum(wd // kw for _, kw in levels) * c * 4
        assert need <= L.PYRAMID_STATIC_MAX_BYTES
        xd = torch.randn(shape, device='cuda', generator=g).to(tdt)
        ref, got = _pool_both(L, xd, levels, code, tdt)
        for r, o, lv in zip(ref, got, levels):
            fill = torch.full((8,), -77.0, dtype=tdt).view(torch.uint8).numpy()
            assert not np.array_equal(o.reshape(-1)[:fill.size], fill), 'output untouched'
            assert np.array_equal(r, o), 'level %s differs between the two forms' % (lv,)


def test_static_scratch_pyramid_pool_rejects_oversize():
    import gpu_util as gu
    from fss import _lib as L
    code, tdt, _ = gu.DT['bf16']
    n, h, wd, c = 16, 33, 65, 2048
    levels = [(int(h / f), int(wd / f)) for f in (1, 2, 3, 4)]
    assert n * h * sum(wd // kw for _, kw in levels) * c * 4 > L.PYRAMID_STATIC_MAX_BYTES
    xd = torch.zeros((n, h, wd, c), dtype=tdt, device='cuda')
    Lv = len(levels)
    outs = [torch.zeros((n, h // kh, wd // kw, c), dtype=tdt, device='cuda') for kh, kw in levels]
    ptrs = (ctypes.c_void_p * Lv)(*[o.data_ptr() for o in outs])
    khs = (ctypes.c_int * Lv)(*[l[0] for l in levels])
    kws = (ctypes.c_int * Lv)(*[l[1] for l in levels])
    bhs = (ctypes.c_int * Lv)(*[h // l[0] for l in levels])
    bws = (ctypes.c_int * Lv)(*[wd // l[1] for l in levels])
    with pytest.raises(L.FssError, match='FSS_PYRAMID_STATIC_MAX_BYTES'):
        L.call('fss_avgpool_pyramid_nhwc_static', xd.data_ptr(), ptrs, khs, kws, bhs, bws, Lv,
               n, h, wd, c, code, None)


def test_static_scratch_pyramid_pool_under_graph_capture():
    """The use the entry point is for: after one warm-up call on the device (which resolves the
    static array's address; fss_set_device does the same), the call is stream-captured into a
    hipGraph, replayed twice, gives the workspace form's bits, and device memory in use does not
    change across capture and replays."""
    import gpu_util as gu
    from fss import _lib as L
    code, tdt, _ = gu.DT['bf16']
    shape, factors = (2, 33, 65, 512), (1, 2, 3, 4)
    n, h, wd, c = shape
    levels = [(int(h / f), int(wd / f)) for f in factors]
    g = torch.Generator(device='cuda').manual_seed(9)
    xd = torch.randn(shape, device='cuda', generator=g).to(tdt)
    ref, got = _pool_both(L, xd, levels, code, tdt)           # also the warm-up call
    Lv = len(levels)
    khs = (ctypes.c_int * Lv)(*[l[0] for l in levels])
    kws = (ctypes.c_int * Lv)(*[l[1] for l in levels])
    bhs = (ctypes.c_int * Lv)(*[h // l[0] for l in levels])
    bws = (ctypes.c_int * Lv)(*[wd // l[1] for l in levels])
    outs = [torch.full((n, h // kh, wd // kw, c), -77.0, dtype=tdt, device='cuda') for kh, kw in levels]
    ptrs = (ctypes.c_void_p * Lv)(*[o.data_ptr() for o in outs])
    stream = torch.cuda.Stream()
    sh = ctypes.c_void_p(stream.cuda_stream)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    L.call('fss_capture_begin', sh)
    ex = ctypes.c_void_p()
    try:
        L.call('fss_avgpool_pyramid_nhwc_static', xd.data_ptr(), ptrs, khs, kws, bhs, bws, Lv, n, h, wd, c, code, sh)
    finally:
        L.call('fss_capture_end', sh, ctypes.byref(ex))
    for _ in range(2):
        for o in outs:
            o.fill_(-77.0)
        torch.cuda.synchronize()
        L.call('fss_graph_launch', ex, sh)
        L.call('fss_stream_sync', sh)
        for r, o in zip(ref, outs):
            assert np.array_equal(r, o.contiguous().view(torch.uint8).cpu().numpy())
    free_after = torch.cuda.mem_get_info()[0]
    L.call('fss_device_sync')
    L.call('fss_graph_destroy', ex)
    # the executable graph itself takes a little driver memory; a 32 MiB scratch allocation would show
    assert free_before - free_after < (8 << 20), (free_before, free_after)
