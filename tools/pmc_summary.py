"""Aggregates rocprofv3 --pmc CSV output (p_counter_collection.csv) into per-kernel means per counter.
Usage: python tools/pmc_summary.py <dir-or-csv> [substring-filter] > summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(path, filt=''):
    files = [path] if path.endswith('.csv') else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                if filt and filt not in name:
                    continue
                c = acc[name[:90]][row['Counter_Name']]
                c[0] += float(row['Counter_Value']); c[1] += 1
    counters = sorted({c for k in acc.values() for c in k})
    print('kernel,dispatches,' + ','.join(counters))
    for name, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        n = max(v[1] for v in cs.values())
        print('"%s",%d,' % (name, n) + ','.join('%.1f' % (cs[c][0] / cs[c][1]) if c in cs else '' for c in counters))


if __name__ == '__main__':
    main(*sys.argv[1:])
