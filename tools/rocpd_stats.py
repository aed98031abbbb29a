"""Summarises a rocprofv3 rocpd database (--kernel-trace) into a per-kernel stats table (like --stats' CSV)."""
import sqlite3
import sys


def main(path, out=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ['name,calls,total_ns,avg_ns,min_ns,max_ns,pct']
    for r in rows:
        lines.append('"%s",%d,%d,%.0f,%d,%d,%.2f' % (r[0][:110], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:])
