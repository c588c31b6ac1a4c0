#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: tools/kstats.py FILE [N]"""
import csv, sys
r = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for x in r[:n]:
    print(x['Name'].split('(')[0][-40:].ljust(40), x['Calls'].rjust(6), ('%.1f' % (float(x['TotalDurationNs']) / 1e3)).rjust(12), 'us  avg',
          ('%.1f' % (float(x['AverageNs']) / 1e3)).rjust(10), 'us', x['Percentage'])
