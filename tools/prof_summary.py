#!/usr/bin/env python
"""rocprofv3 (--kernel-trace --stats, rocpd sqlite output) -> plain-text per-kernel summary for profiles/.

    python tools/prof_summary.py gpurun_out/prof2/r2_results.db [steps] > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels "
                       "group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f'# rocprofv3 --kernel-trace --stats summary: {n} dispatches, {tot / 1e6:.3f} ms of kernel time'
          + (f', {steps} bench steps (+warm-up/capture) -> {tot / 1e6 / steps:.3f} ms kernel time per step' if steps else ''))
    print(f'{"%time":>6} {"calls":>7} {"avg_us":>9} {"min_us":>8} {"max_us":>8} {"vgpr":>5} {"agpr":>5} {"lds":>7} {"scratch":>7}  kernel')
    for r in rows:
        name = re.sub(r'\(anonymous namespace\)::|^void ', '', r[0])
        print(f'{r[2] / tot * 100:6.2f} {r[1]:7d} {r[3] / 1e3:9.2f} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:7d}  {name[:150]}')


if __name__ == '__main__':
    main()
