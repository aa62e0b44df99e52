"""Per-layer counter summary of the conv kernel from the SQ / TCP / TD passes of
tools/profile_round.sh: <out>/<tag>_conv_pmc.txt.

bench.py ends with eager single-stream passes over the plan's launches (CompiledGraph.profile);
the last `launches_per_step` conv_igemm_kernel / conv8_kernel dispatches of every counter run are the conv
layers of the final pass, in the order of <tag>_layers.txt.  Shown for the layers that take the
most time: MFMA pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (per-SIMD busy
cycles over CU busy cycles x 4 SIMDs is how the hardware guide normalises it), LDS wait share,
TCP pending-stall and texture-data busy shares of the kernel's GRBM_GUI_ACTIVE cycles.
"""
import collections
import csv
import glob
import json
import os
import sys


def conv_rows(dirname, n):
    path = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)[0]
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if 'conv_igemm_kernel' in r['Kernel_Name'] or 'conv8_kernel' in r['Kernel_Name']:
                per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    out = {}
    for name, rows in per.items():
        rows.sort()
        assert len(rows) >= n, (name, len(rows), n)
        out[name] = [v for _, v in rows[-n:]]
    return out


def main():
    out, tag = sys.argv[1], sys.argv[2]
    with open(os.path.join(out, tag + '_bench.json')) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    n = bench['roofline']['launches_per_step']
    layers = []
    with open(os.path.join(out, tag + '_layers.txt')) as f:
        for line in f:
            p = line.split()
            if p and p[0] == 'conv':
                i = p.index('ms')
                layers.append((p[1], float(p[i - 1]), float(p[i + 1]), float(p[i + 3]), p[-1]))
    assert len(layers) == n, (len(layers), n)
    c = {}
    for d in sorted(glob.glob(os.path.join(out, tag + '_pmc_g*'))):
        if os.path.isdir(d):
            c.update(conv_rows(d, n))
    order = sorted(range(n), key=lambda k: -layers[k][1])
    lines = ['# conv_igemm_kernel (configuration 35: conv8_kernel), per layer (the %d slowest of %d), %s, %d frames per launch' % (
        min(16, n), n, bench['dtype'], bench['config']['frames_per_step_per_gpu']),
        '# mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); lds_wait = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES;',
        '# tcp_stall = TCP_PENDING_STALL_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs); td_busy = TD_TD_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs)',
        '# mfma_issue = SQ_INSTS_MFMA * (16 cycles per v_mfma_*_16x16x32, 32 per 32x32x16) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)',
        '%-62s %8s %8s %8s %6s %10s %8s %9s %8s %s' % ('layer', 'ms', 'TFLOP/s', 'GB/s', 'mfma', 'mfma_issue', 'lds_wait', 'tcp_stall', 'td_busy', 'config')]
    # 16 cycles per v_mfma_*_16x16x32 (bf16 / f16; the split-bf16 launches issue three per product),
    # 32 per 32x32 shape (the exact float32 mode, or FSS_MFMA_EDGE=32)
    mfma_cycles = 32.0 if bench['config'].get('compute_mode') == 'float32' or os.environ.get('FSS_MFMA_EDGE') == '32' else 16.0
    cus = 256.0 / 8.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs

    def ratio(a, b, k, scale=1.0):
        if a in c and b in c and c[b][k] > 0:
            return '%.2f' % (c[a][k] / (c[b][k] * scale))
        return 'n/a'

    for k in order[:16]:
        name, ms, tf, gb, cfg = layers[k]
        issue = 'n/a'
        if 'SQ_INSTS_MFMA' in c and c.get('GRBM_GUI_ACTIVE', [0] * n)[k] > 0:
            issue = '%.2f' % (c['SQ_INSTS_MFMA'][k] * mfma_cycles / (c['GRBM_GUI_ACTIVE'][k] / 8.0 * 1024.0))
        lines.append('%-62s %8.4f %8.1f %8.1f %6s %10s %8s %9s %8s %s' % (
            name[-62:], ms, tf, gb,
            ratio('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', k, 4.0), issue,
            ratio('SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES', k),
            ratio('TCP_PENDING_STALL_CYCLES_sum', 'GRBM_GUI_ACTIVE', k, cus),
            ratio('TD_TD_BUSY_sum', 'GRBM_GUI_ACTIVE', k, cus), cfg))
    tot = {name: sum(v) for name, v in c.items()}
    lines.append('# sums over all %d conv launches: %s' % (n, ', '.join('%s=%.4g' % kv for kv in sorted(tot.items()))))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in tot and tot.get('SQ_BUSY_CU_CYCLES'):
        lines.append('# whole-step MFMA pipe busy share of the conv launches: %.3f'
                     % (tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * tot['SQ_BUSY_CU_CYCLES'])))
    with open(os.path.join(out, tag + '_conv_pmc.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
