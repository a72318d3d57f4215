"""Driver for the HBM-traffic PMC passes: launches the five kernels of a DiT block through their C-ABI op entry points, eagerly,
in the order qkv, attention, out-proj, FF1, FF2 (the same closures and shapes bench.py times), so that a
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass attributes bytes per launch to each of them -- out-proj and FF2 share
one kernel symbol and are told apart by their position in the dispatch order (tools/pmc_block_ops_summary.py).
usage: python tools/pmc_block_ops.py <precision> <batch>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LAUNCHES = 6


def eager_launches(fn, dev, iters):
    for _ in range(LAUNCHES):
        fn()
    torch.cuda.synchronize()
    return 1.0


def main():
    precision, batch = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0")
    bench._time_launches = eager_launches
    out, _ = bench.kernel_rooflines(precision, dev, batch)
    for k in out:
        print(k["key"], k["shape"], k["algorithmic_bytes"])


if __name__ == "__main__":
    main()
