"""Per-kernel averages of rocprofv3 --pmc passes (csv output): one *_counter_collection.csv per pass.
python tools/pmc_summary.py KERNEL_SUBSTRING DIR [DIR ...]"""
import collections
import csv
import glob
import sys

pat = sys.argv[1]
for i, d in enumerate(sys.argv[2:], 1):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value'])
                cnt[r['Counter_Name']] += 1
    print('pass %d:' % i)
    for k in sorted(acc):
        print('  %-28s %.6g   (n=%d)' % (k, acc[k] / cnt[k], cnt[k]))
