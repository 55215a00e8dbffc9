"""MFMA-pipe utilisation and wave wait share per kernel from one rocprofv3 pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_BUSY_CYCLES SQ_WAVES) of tools/kernel_bench.py.

    mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDS * SIMDS)
(the derived counter MfmaUtil of rocprofv3, written out: the per-dispatch GRBM_GUI_ACTIVE of the CSV is
summed over the 8 XCDs of an MI355X - checked against the kernel duration x clock - and the chip has
256 CUs x 4 SIMDs; one v_mfma_f32_16x16x4_f32 keeps its SIMD's pipe busy for 32 cycles.)

usage: python tools/pmc_mfma.py <counter_collection.csv> [out.json]
"""
import collections
import csv
import json
import re
import sys

XCDS, SIMDS = 8, 1024
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if 'feta::' not in name:
        continue
    short = re.sub(r'\(.*', '', name).replace('void ', '')
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k in sorted(acc):
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    if m.get('GRBM_GUI_ACTIVE', 0) <= 0:
        continue
    cyc = m['GRBM_GUI_ACTIVE'] / XCDS
    out[k] = {
        'gpu_cycles': round(cyc),
        'mfma_busy_pct': round(100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * SIMDS), 2),
        'wait_any_pct_of_wave_cycles': round(100.0 * m.get('SQ_WAIT_ANY', 0.0) / max(m.get('SQ_WAVE_CYCLES', 1.0), 1.0), 1),
        'mfma_insts': round(m.get('SQ_INSTS_MFMA', 0.0)),
        'valu_insts': round(m.get('SQ_INSTS_VALU', 0.0)),
        'waves': round(m.get('SQ_WAVES', 0.0)),
    }
    print('%-56s mfma busy %5.1f %%   waves parked %4.1f %%' % (k[:56], out[k]['mfma_busy_pct'],
                                                             out[k]['wait_any_pct_of_wave_cycles']))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
