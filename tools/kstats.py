"""Per-step summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv.
usage: python tools/kstats.py <kernel_stats.csv> <steps-run-by-the-command> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print('kernel time per step: %.1f us, launches per step: %.1f' % (tot / 1e3 / steps, calls / steps))
for r in rows[:top]:
    print('%-78s n/step=%6.1f us/step=%8.1f avg=%7.2fus %5.1f%%' % (
        r['Name'][:78], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e3 / steps,
        float(r['AverageNs']) / 1e3, float(r['Percentage'])))
