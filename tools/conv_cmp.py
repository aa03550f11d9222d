"""Side-by-side view of two or more tools/conv_bench.py tables (per-layer ms).  python tools/conv_cmp.py A.txt B.txt ..."""
import re
import sys


def load(f):
    d = {}
    for l in open(f):
        m = re.match(r'\s*([\d.]+)ms (\w+)\s+x=(\(.*?\))\s+w=(\(.*?\))\s+s(\d) u(\d) n=\s*(\d+)\s+([\d.]+)ms', l)
        if m:
            d[(m.group(2), m.group(3), m.group(4), m.group(5))] = (float(m.group(8)), int(m.group(7)))
    return d


tabs = [load(f) for f in sys.argv[1:]]
kinds = sys.argv[1:]
for k in tabs[0]:
    print('%-6s %-22s %-20s s%s n=%2d ' % (k[0], k[1], k[2], k[3], tabs[0][k][1]) +
          ' '.join('%7.3f' % t.get(k, (float('nan'),))[0] for t in tabs))
for kind in ('fwd', 'dgrad', 'wgrad'):
    print(kind, ' '.join('%7.2f' % sum(v[0] * v[1] for k, v in t.items() if k[0] == kind and k in tabs[0]) for t in tabs))
