"""Per-kernel totals from a rocprofv3 kernel_stats CSV: python tools/kstats.py <dir> [pattern]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
if not f:
    print('no kernel_stats.csv under', sys.argv[1]); sys.exit(1)
for row in csv.DictReader(open(f[0])):
    if pat in row['Name']:
        print('%-60s calls %4s  total %9.3f ms  avg %8.1f us  max %8.1f us' % (row['Name'].replace('(anonymous namespace)::', '')[:60], row['Calls'], float(row['TotalDurationNs']) / 1e6,
                                                                              float(row['AverageNs']) / 1e3, float(row['MaxNs']) / 1e3))
