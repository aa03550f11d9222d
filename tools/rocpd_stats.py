"""Per-kernel summary (calls, total, average, share) of a rocprofv3 run kept in its rocpd SQLite form
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes DIR/NAME_results.db), written as the same CSV columns the
csv output format uses.   python tools/rocpd_stats.py RESULTS.db [OUT.csv]"""
import csv
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute("select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     "from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    return [dict(Name=r[0], Calls=r[1], TotalDurationNs=int(r[2]), AverageNs=float(r[3]),
                 Percentage=100.0 * r[2] / total, MinNs=int(r[4]), MaxNs=int(r[5])) for r in rows]


if __name__ == '__main__':
    st = kernel_stats(sys.argv[1])
    if len(sys.argv) > 2:
        with open(sys.argv[2], 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(st[0].keys()))
            w.writeheader()
            w.writerows(st)
    tot = sum(r['TotalDurationNs'] for r in st)
    print('total kernel time %.2f ms over %d kernels' % (tot / 1e6, len(st)))
    for r in st[:50]:
        print('%-100s calls %6d tot %8.2f ms avg %9.1f us %5.2f%%' % (r['Name'][:100], r['Calls'],
              r['TotalDurationNs'] / 1e6, r['AverageNs'] / 1e3, r['Percentage']))
