"""Prints the per-parameter errors of the bf16 storage leg against the fp64 oracle (what tests/bench_checks.py bounds)."""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench_checks as BC      # noqa: E402

if __name__ == '__main__':
    errs, used = BC.check_bench_step(torch.device('cuda:0'), contextlib.nullcontext, sys.argv[1:] + ['--dtype', 'bf16'],
                                     replays=2)
    for k, v in errs.items():
        print('%-45s %.3e' % (k, v))
