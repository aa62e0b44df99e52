"""Re-creates the packed-f32 fault: pyramid_add_kernel gives wrong elements when the side chain
(headed by the uint8 stem convolution) starts right before it, if fss_kernels.hip was built WITH
hipcc's SLP vectorizer, and characterises the wrong elements: which chunk element, which lanes /
waves, and which arithmetic slip reproduces the stored value.

    # library variant with packed math (only fss_kernels.hip differs from the shipped build):
    cd fast-semantic-segmentation_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
        -I../../include -c fss_kernels.hip -o /tmp/k_slp.o && hipcc --offload-arch=gfx950 -shared \
        -fPIC -o ../../exp/libfss_slp.so fss_runtime.o fss_conv.o fss_stem.o fss_stem2.o /tmp/k_slp.o
    python tools/anomaly_packed_f32.py slp nostem2 nodual     # faulty: ~2000 wrong elements / pass
    python tools/anomaly_packed_f32.py nostem2 nodual         # shipped build: none
"""
import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'slp':
    os.environ['FSS_LIB'] = os.path.join(ROOT, 'exp', 'libfss_slp.so')
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from test_models_gpu import build
from fss import plan, _lib
print('library', _lib.LIB_PATH, flush=True)
CASE = ('icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0])
PAD = (1025, 2049)
NB = 2
rng = np.random.default_rng(12)
frames = rng.integers(0, 256, (NB, 1024, 2048, 3)).astype(np.uint8)
g, cfg, outs, weights = build(*CASE, (NB, 1024, 2048, 3), PAD)
cg = plan.CompiledGraph(g, outs, weights, dtype='bfloat16', streams=1, commute_zoom=False,
                        fuse_stem2='nostem2' not in sys.argv, fuse_dual='nodual' not in sys.argv, fuse_head='nodual' not in sys.argv)
cg.run(frames)
i_pyr = [i for i, s in enumerate(cg.steps) if s.kind == 'pyramid'][0]
i_avg = [i for i, s in enumerate(cg.steps) if s.kind == 'avgpool'][0]
rel = i_avg - 1
print('steps', [(i, s.kind, s.name[-28:]) for i, s in enumerate(cg.steps) if i >= rel - 1][:12], flush=True)

def buf_at(ptr):
    for v in list(cg.vals.values()) + cg.temps:
        bp = v.buf.data_ptr()
        if bp <= ptr < bp + v.buf.numel() * v.buf.element_size():
            return v.buf
    raise KeyError(ptr)

s_pyr = cg.steps[i_pyr]
x_buf = buf_at(s_pyr.args[0])
y_buf = buf_at(s_pyr.args[5])
lv_bufs = [buf_at(q) for q in s_pyr.args[1]]
torch.cuda.synchronize()
ref_y = y_buf.clone()
x = x_buf.float().cpu().numpy()
lv = [b.float().cpu().numpy() for b in lv_bufs]
N, H, W, C = x.shape

def coord(o, insz, outsz):
    sc = np.float32(insz - 1) / np.float32(outsz - 1) if outsz > 1 else np.float32(0)
    src = np.float32(o) * sc
    lo = min(int(np.floor(src)), insz - 1)
    hi = min(lo + 1, insz - 1)
    return lo, hi, np.float32(src - np.floor(src))

def terms(n, h, w, c):
    """per level: (v0, dv, lx, term) at element c, as the kernel computes them (float32)."""
    out = []
    for p in lv:
        bh, bw = p.shape[1:3]
        y0, y1, ly = coord(h, bh, H)
        x0, x1, lx = coord(w, bw, W)
        tl, bl, tr, br = (np.float32(p[n, y0, x0, c]), np.float32(p[n, y1, x0, c]),
                          np.float32(p[n, y0, x1, c]), np.float32(p[n, y1, x1, c]))
        v0 = np.float32(tl + (bl - tl) * ly)
        dv = np.float32(np.float32(tr + (br - tr) * ly) - v0)
        out.append((v0, dv, lx, np.float32(v0 + dv * lx)))
    return out

def bf16(v):
    t = torch.tensor([float(v)], dtype=torch.float32).to(torch.bfloat16)
    return float(t[0])

def analyse(tag, replays):
    for it in range(replays):
        cg.launch(); cg.sync()
        now = y_buf
        d = (now != ref_y).nonzero()
        print(tag, 'iter', it, 'differing elements', d.shape[0], flush=True)
        if d.shape[0] == 0:
            continue
        dd = d.cpu().numpy()
        print('  element within chunk (c%8):', sorted(collections.Counter((dd[:, 3] % 8).tolist()).items()))
        px = collections.defaultdict(list)
        for n, h, w, c in dd.tolist():
            px[(n, h, w)].append(c // 8)
        print('  pixels hit', len(px), ' w%8 (pixel slot in the block):',
              sorted(collections.Counter([k[2] % 8 for k in px]).items()))
        for k, chunks in list(px.items())[:6]:
            waves = sorted(collections.Counter([q // 64 for q in chunks]).items())
            print('   pixel', k, 'chunks', len(chunks), 'range', min(chunks), max(chunks), 'waves', waves)
        nowc = now.float().cpu().numpy()
        refc = ref_y.float().cpu().numpy()
        hits = collections.Counter()
        for n, h, w, c in dd[:400].tolist():
            t = terms(n, h, w, c)
            cp = c ^ 1                         # the other element of the packed pair
            tp = terms(n, h, w, cp)
            full = np.float32(x[n, h, w, c])
            for q in t:
                full = np.float32(full + q[3])
            got = nowc[n, h, w, c]
            label = 'unexplained'
            if bf16(full) != refc[n, h, w, c]:
                label = 'cpu model disagrees with ref'
            else:
                for L in range(len(lv)):
                    # (1) level L's term missing
                    alt = np.float32(x[n, h, w, c])
                    for j, q in enumerate(t):
                        if j != L:
                            alt = np.float32(alt + q[3])
                    if bf16(alt) == got:
                        label = 'level %d term missing' % L; break
                    # (2) high half multiplied the pair partner's RESULT instead of lx
                    alt = np.float32(x[n, h, w, c])
                    for j, q in enumerate(t):
                        alt = np.float32(alt + (np.float32(t[L][0] + t[L][1] * tp[L][3]) if j == L else q[3]))
                    if bf16(alt) == got:
                        label = 'level %d: lx replaced by partner result' % L; break
                    # (3) the partner's term was added instead
                    alt = np.float32(x[n, h, w, c])
                    for j, q in enumerate(t):
                        alt = np.float32(alt + (tp[L][3] if j == L else q[3]))
                    if bf16(alt) == got:
                        label = 'level %d: partner term added' % L; break
            hits[label] += 1
        print('  explanation of the first 400:', dict(hits), flush=True)
        n, h, w, c = dd[0].tolist()
        print('  example', (n, h, w, c), 'now', nowc[n, h, w, c], 'ref', refc[n, h, w, c], 'x', x[n, h, w, c],
              'terms', [float(q[3]) for q in terms(n, h, w, c)], 'partner terms', [float(q[3]) for q in terms(n, h, w, c ^ 1)])

cg._setup_streams(tune=False)
cg._assign_streams(rel)
cg.two_streams = True
cg._capture()
analyse('graph, fork after step %d' % rel, 4)
# the same schedule launched eagerly (no graph)
ge, cg.graph_exec = cg.graph_exec, None
analyse('eager two streams', 4)
cg.graph_exec = ge
# fork at the start (the shipped schedule)
cg._assign_streams(None)
cg._capture()
analyse('graph, fork at start', 4)
