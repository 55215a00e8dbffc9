#!/usr/bin/env python3
"""Rounds of resident workgroups per launch of a captured step (rocprofv3 kernel trace of bench.py).  A launch whose grid is a
little more than a whole number of rounds pays a full extra round for a few workgroups (EXPERIMENTS.md, round 4: config 4).
slots = 256 CUs x workgroups per CU BY REGISTERS (512 per SIMD lane, waves per SIMD capped at 8; the trace's VGPR_Count is
half the wave64 allocation).  Dynamic LDS is not in the trace (LDS_Block_Size is the static part): kernels that take most of a
CU's 160 KB - the attention blocks, attn_out_fwd, attn_bwd_head, coeff_fwd_wide - hold ONE workgroup per CU whatever this
prints; csrc/*_lds_bytes() is the authority there.
usage: tools/rounds.py <kernel_trace.csv>"""
import csv, sys, statistics as st

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] + '/' + r['Grid_Size_X'] for r in rows]
mid = len(rows) // 2
per = None
for p in range(4, 200):
    if all(names[mid + i] == names[mid + i - p] for i in range(p, 400)):
        per = p
        break
assert per, 'no period found'
start = mid
while start - per >= 0 and names[start - per:start] == names[start:start + per]:
    start -= per
gaps = [int(rows[start + j]['Start_Timestamp']) - int(rows[start + j - 1]['End_Timestamp']) for j in range(per)]
start += max(range(per), key=lambda j: gaps[j])
for j in range(per):
    r = rows[start + j]
    dur = st.median(int(rows[start + j + k * per]['End_Timestamp']) - int(rows[start + j + k * per]['Start_Timestamp'])
                    for k in range(10) if start + j + (k + 1) * per < len(rows)) / 1e3
    wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1)
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) // wg
    regs = 2 * int(r['VGPR_Count']) + int(r.get('Accum_VGPR_Count', 0) or 0)
    lds = int(r['LDS_Block_Size'])
    waves_simd = max(1, min(8, 512 // max(regs, 1)))
    by_regs = max(1, 4 * waves_simd // max(1, wg // 64))
    by_lds = 160 * 1024 // lds if lds > 0 else 99
    per_cu = max(1, min(by_regs, by_lds))
    slots = 256 * per_cu
    print('%2d %-52s grid %5d x %4d thr  regs %3d lds %6d  -> %d / CU, %5d slots, %5.2f rounds  %6.1f us' % (
        j, r['Kernel_Name'][:52], grid, wg, regs, lds, per_cu, slots, grid / slots, dur))
