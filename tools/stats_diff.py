"""Compare two per-kernel CSV summaries (tools/rocpd_stats.py / rocprofv3 --stats columns): per kernel name the
change in calls and total time.   python tools/stats_diff.py OLD.csv NEW.csv [steps]"""
import csv
import sys


def load(p):
    with open(p) as f:
        return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(f)}


a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0))
    cb, tb = b.get(k, (0, 0))
    rows.append((tb - ta, k, ca, cb, ta, tb))
rows.sort(key=lambda r: -abs(r[0]))
print('total: %.3f -> %.3f ms/step' % (sum(v[1] for v in a.values()) / 1e6 / steps, sum(v[1] for v in b.values()) / 1e6 / steps))
for d, k, ca, cb, ta, tb in rows[:40]:
    print('%+9.3f ms/step  calls %7.1f -> %7.1f  %9.3f -> %9.3f  %s' % (d / 1e6 / steps, ca / steps, cb / steps, ta / 1e6 / steps, tb / 1e6 / steps, k[:110]))
