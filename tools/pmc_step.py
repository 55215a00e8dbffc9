"""Per-kernel HBM traffic and MFMA / wait counters of ONE bench.py configuration, from three rocprofv3 --pmc passes of the
bench itself (tools/profile_config.sh: FETCH_SIZE, WRITE_SIZE, and the SQ / GRBM pass; eager launches, --no-graph):

    python tools/pmc_step.py <fetch.csv> <write.csv> <sq.csv> [out.json]

Every launch of a kernel symbol in the run is averaged.  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950
FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section; calibrated on bn_apply_fwd in round 1).
mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (tools/pmc_mfma.py)."""
import collections
import csv
import json
import re
import sys

XCDS, SIMDS = 8, 1024


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '')


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write, sq = (per_kernel(p) for p in sys.argv[1:4])
    out = {}
    for k in sorted(set(fetch) | set(write) | set(sq)):
        row = {}
        f, w = fetch.get(k, {}).get('FETCH_SIZE'), write.get(k, {}).get('WRITE_SIZE')
        if f and w:
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            row.update(launches=len(f), FETCH_SIZE_KB=round(fk, 1), WRITE_SIZE_KB=round(wk, 1),
                       hbm_bytes=int(round((2.0 * fk + wk) * 1024)))
        m = {c: sum(v) / len(v) for c, v in sq.get(k, {}).items()}
        if m.get('GRBM_GUI_ACTIVE', 0) > 0:
            cyc = m['GRBM_GUI_ACTIVE'] / XCDS
            row.update(gpu_cycles=round(cyc),
                       mfma_busy_pct=round(100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * SIMDS), 2),
                       wait_any_pct_of_wave_cycles=round(100.0 * m.get('SQ_WAIT_ANY', 0.0) / max(m.get('SQ_WAVE_CYCLES', 1.0), 1.0), 1),
                       mfma_insts=round(m.get('SQ_INSTS_MFMA', 0.0)), valu_insts=round(m.get('SQ_INSTS_VALU', 0.0)),
                       waves=round(m.get('SQ_WAVES', 0.0)))
        if row:
            out[k] = row
    for k, r in out.items():
        if 'feta::' in k or 'Cijk' in k:
            print('%-64s %8.2f MB  mfma %5s %%  wait %5s %%' % (k[:64], r.get('hbm_bytes', 0) / 1e6, r.get('mfma_busy_pct'),
                                                                r.get('wait_any_pct_of_wave_cycles')))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
