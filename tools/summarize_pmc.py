"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into
<out>/<tag>_conv_traffic.json (HBM bytes of the conv kernel per bench step).

The last `launches_per_step` conv dispatches of each pass are the final eager
profiling pass of bench.py (one launch per conv layer of the plan).  TCC counters
are in KiB; FETCH_SIZE is doubled: on gfx950 it counts the 128-byte requests
that 16-byte-per-lane streams generate as 64 bytes.
"""
import csv
import glob
import json
import os
import sys


def conv_kb(dirname, counter, n):
    path = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter and ('conv_igemm_kernel' in r['Kernel_Name'] or 'conv8_kernel' in r['Kernel_Name']):
                rows.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    rows.sort()
    assert len(rows) >= n, (len(rows), n)
    return sum(v for _, v in rows[-n:])


def main():
    out, tag = sys.argv[1], sys.argv[2]
    with open(os.path.join(out, tag + '_bench.json')) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    n = bench['roofline']['launches_per_step']
    batch = bench['config']['frames_per_step_per_gpu']
    fetch = conv_kb(os.path.join(out, tag + '_pmc_fetch'), 'FETCH_SIZE', n)
    write = conv_kb(os.path.join(out, tag + '_pmc_write'), 'WRITE_SIZE', n)
    total = (2.0 * fetch + write) * 1024.0
    d = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py, '
                   'last eager profiling pass',
         'dtype': bench['config'].get('compute_mode') or ('bfloat16' if bench['dtype'] == 'bf16' else 'float32'),
         'frames_per_step': batch, 'conv_launches_per_step': n,
         'conv_fetch_size_kb_raw': fetch, 'conv_write_size_kb_raw': write,
         'fetch_correction': 'x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B)',
         'conv_hbm_bytes_per_step': total, 'conv_hbm_bytes_per_launch_avg': total / n,
         'conv_algorithmic_bytes_per_launch_avg': bench['roofline'].get('algorithmic_bytes')}
    with open(os.path.join(out, tag + '_conv_traffic.json'), 'w') as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d))


if __name__ == '__main__':
    main()
