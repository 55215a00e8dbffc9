#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel trace of `bench.py` (hipGraph replays): finds the steady-state period of the
kernel sequence, then prints the median span of a step, the median busy time (sum of kernel durations) and, per kernel
position in the step, name / grid / median duration / median gap to the previous kernel.
usage: tools/trace_step.py <kernel_trace.csv> [launches-per-step]"""
import csv, sys, statistics as st

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Grid_Size_X']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
names = [r[2] + '/' + r[3] for r in rows]
per = int(sys.argv[2]) if len(sys.argv) > 2 else None
mid = len(rows) // 2
if per is None:
    win = names[mid:mid + 400]
    for p in range(4, 200):
        if all(win[i] == win[i - p] for i in range(p, len(win))):
            per = p
            break
assert per, 'no period found'
# grow the periodic region around the middle of the trace
start = mid
while start - per >= 0 and names[start - per:start] == names[start:start + per]:
    start -= per
end = start + per
while end + per <= len(rows) and names[end:end + per] == names[start:start + per]:
    end += per
# align the step boundary on the largest gap inside a period
gaps = [rows[start + j][0] - rows[start + j - 1][1] if j else rows[start + per][0] - rows[start + per - 1][1] for j in range(per)]
off = max(range(per), key=lambda j: gaps[j])
start += off
end -= per - off if off else 0
steps = [(rows[i:i + per]) for i in range(start, end, per)]
steps = steps[len(steps) // 4:]         # skip warm-up replays
span = [s[-1][1] - s[0][0] for s in steps]
busy = [sum(k[1] - k[0] for k in s) for s in steps]
period = [steps[i + 1][0][0] - steps[i][0][0] for i in range(len(steps) - 1)]
print('%d launches per step, %d steps: period %.1f us, span %.1f us, busy %.1f us' % (per, len(steps), st.median(period) / 1e3, st.median(span) / 1e3, st.median(busy) / 1e3))
for j in range(per):
    dur = st.median(s[j][1] - s[j][0] for s in steps) / 1e3
    gap = st.median((s[j][0] - s[j - 1][1]) for s in steps) / 1e3 if j else 0.0
    print('%2d %-62s %8s %7.1f us  gap %5.1f' % (j, steps[0][j][2][:62], steps[0][j][3], dur, gap))
