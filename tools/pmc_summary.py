"""Joins rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, counter unit = KB) of
tools/kernel_bench.py into HBM traffic per kernel configuration.  The dispatch order of the
micro-benchmark is deterministic (3 warm-up + ITERS calls per row); rows are matched by the kernel
symbols each C-ABI call enqueues.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> ITERS [out.json]
Correction (MI355X_MICROARCH.md, HBM section; calibrated here on bn_apply_fwd, a pure float4 stream:
74.5 MiB counted for 148 MiB read): FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE exact.
"""
import csv
import json
import sys

ROWS = [  # (bench row, kernel-symbol substrings enqueued by one call, in order)
    ('attn_fwd (+attn write)', ['attn_fwd']),
    ('attn_fwd (no attn write)', ['attn_fwd']),
    ('attn_bwd (dq + dkdv)', ['attn_bwd_dq', 'attn_bwd_dkdv']),
    ('coeff_fwd', ['coeff_fwd_kernel']),
    ('coeff_bwd', ['coeff_bwd_kernel', 'colsum']),
    ('spec_filter_fwd', ['spec_fwd']),
    ('spec_filter_bwd', ['spec_bwd']),
    ('cheb_filter_fwd', ['cheb_fwd']),
    ('cheb_filter_bwd', ['cheb_bwd']),
    ('rowlin_fwd in_proj', ['rowlin_fwd']), ('rowlin_bwd in_proj', ['rowlin_bwd', 'colsum']),
    ('rowlin_fwd out_proj', ['rowlin_fwd']), ('rowlin_bwd out_proj', ['rowlin_bwd', 'colsum']),
    ('rowlin_fwd linear1', ['rowlin_fwd']), ('rowlin_bwd linear1', ['rowlin_bwd', 'colsum']),
    ('rowlin_fwd linear2', ['rowlin_fwd']), ('rowlin_bwd linear2', ['rowlin_bwd', 'colsum']),
    ('bn_apply_fwd', ['bn_apply_fwd']),
    ('bn_bwd (reduce + apply)', ['bn_bwd_reduce', 'bn_bwd_apply']),
    # feta_tmlr_amd/benchcases.py: the kernels of one fused-stack layer, in the stack's variants
    ('attn_block_fwd (no attn write)', ['attn_block_fwd']),
    ('attn_block_fwd (+attn write)', ['attn_block_fwd']),
    ('ffn_fwd', ['ffn_fwd']),
    ('ffn_fwd (+ coefficient generator)', ['ffn_fwd']),
    ('ffn_bwd', ['ffn_bwd']),
    ('ffn_bwd (+ coefficient generator)', ['ffn_bwd']),
    ('ffn_bwd (gradient in two parts)', ['ffn_bwd']),
    ('attn_block_bwd', ['attn_block_bwd']),
    ('attn_block_bwd (two workgroups per graph)', ['attn_block_bwd']),
    ('rowlin_bwd linear_cat', ['rowlin_bwd']),
    ('lin_fwd', ['lin_fwd']),
    ('lin_bwd', ['lin_bwd']),
]


def seq(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [(r['Kernel_Name'], float(r['Counter_Value'])) for r in rows if 'feta::' in r['Kernel_Name']]


def walk(stream, calls):
    out, i = {}, 0
    for name, syms in ROWS:
        tot = 0.0
        for _ in range(calls):
            for s in syms:
                while s not in stream[i][0]:   # e.g. the one-off bn_stats launch of the set-up
                    i += 1
                tot += stream[i][1]
                i += 1
        out[name] = tot / calls
    return out


def main():
    f, w = seq(sys.argv[1]), seq(sys.argv[2])
    if any('attn_bwd_dense' in name for name, _ in f):   # N <= 64: dq and dk/dv roles of ONE launch
        for i, (row, _) in enumerate(ROWS):
            if row.startswith('attn_bwd'):
                ROWS[i] = (row, ['attn_bwd_dense'])
    if any('attn_bwd_graph' in name for name, _ in f):   # large batches: one workgroup per graph
        for i, (row, _) in enumerate(ROWS):
            if row.startswith('attn_bwd'):
                ROWS[i] = (row, ['attn_bwd_graph'])
    if any('attn_block_bwd' in name for name, _ in f) and not any('Lb1ELb0' in name or ', true, false>' in name
                                                                   for name, _ in f):
        # beyond 256 graphs with the fused attention-block backward (bf16): no two-workgroup form, no two-part gradient
        names = [r for r, _ in ROWS]
        del ROWS[names.index('attn_block_bwd (two workgroups per graph)')]
        del ROWS[[r for r, _ in ROWS].index('ffn_bwd (gradient in two parts)')]
    if not any('attn_block_bwd' in name for name, _ in f):   # batches beyond the fused attention-block backward
        i = [r for r, _ in ROWS].index('attn_block_bwd')
        del ROWS[i + 1]      # the two-workgroup form and the FFN backward that takes its two parts: not issued there
        del ROWS[i - 1]
        i -= 1
        ROWS[i:i + 1] = [('rowlin_bwd out_proj (stack: BN-backward gradient)', ['rowlin_bwd']),
                         ('rowlin_bwd in_proj (stack: add, sums)', ['rowlin_bwd'])]
    calls = int(sys.argv[3]) + 3
    fk, wk = walk(f, calls), walk(w, calls)
    res = {name: {'FETCH_SIZE_KB': round(fk[name], 1), 'WRITE_SIZE_KB': round(wk[name], 1),
                  'hbm_bytes': int(round((2.0 * fk[name] + wk[name]) * 1024))} for name, _ in ROWS}
    txt = json.dumps(res, indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
