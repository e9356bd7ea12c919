#!/usr/bin/env python
"""rocprofv3 --pmc CSVs of tools/pmc_step.sh -> per-kernel HBM-side traffic per launch (JSON + text).

    python tools/pmc_step_summary.py gpurun_out/pmcstep profiles/r01_pmc_step_traffic [profiles/r01_bench_kernel_stats_v7.txt]

With the optional rocprofv3 --stats summary (un-profiled durations of the same kernels in the replayed graph) the table also
gives the HBM-side GB/s of every kernel and, from the `mfma` pass, the matrix-core occupancy: SQ_VALU_MFMA_BUSY_CYCLES (cycles,
summed over the 1024 SIMDs) / 1024 / (avg_us x CLOCK_MHZ).  GRBM_GUI_ACTIVE of a profiled dispatch is not used as the time base:
it is accumulated per XCD and the profiled launch is several times longer than the same kernel inside the replayed graph.

FETCH_SIZE / WRITE_SIZE are reported in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies the 128-byte
requests of wide coalesced streaming reads at 64 bytes, so `fetch_bytes_corrected` = 2 x raw; WRITE_SIZE is uncalibrated and
given raw."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

N_SIMD = 256 * 4
CLOCK_MHZ = 2100.0        # sustained shader clock of these kernels (1.9 - 2.3 GHz by DVFS, MI355X_MICROARCH.md)


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = re.sub(r'\(anonymous namespace\)::|^void ', '', row['Kernel_Name'])
            name = re.sub(r'\(.*$', '', name)
            a = agg[name]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch = load(f'{src}/fetch/p_counter_collection.csv', 'FETCH_SIZE')
    write = load(f'{src}/write/p_counter_collection.csv', 'WRITE_SIZE')
    hit = load(f'{src}/tcc/p_counter_collection.csv', 'TCC_HIT_sum')
    miss = load(f'{src}/tcc/p_counter_collection.csv', 'TCC_MISS_sum')
    req = load(f'{src}/tcc/p_counter_collection.csv', 'TCC_REQ_sum')
    mfma = gui = None
    if os.path.exists(f'{src}/mfma/p_counter_collection.csv'):
        mfma = load(f'{src}/mfma/p_counter_collection.csv', 'SQ_VALU_MFMA_BUSY_CYCLES')
        gui = load(f'{src}/mfma/p_counter_collection.csv', 'GRBM_GUI_ACTIVE')
    dur = {}
    if len(sys.argv) > 3:                                   # "%time calls avg_us ... kernel" rows of tools/prof_summary.py
        for line in open(sys.argv[3]):
            f = line.split(None, 9)
            if len(f) == 10 and f[0][0].isdigit():
                dur[re.sub(r'\(.*$', '', f[9].strip())] = float(f[2])
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n = fetch[k][0]
        f_kb = fetch[k][1] / n
        w_kb = write[k][1] / write[k][0] if k in write and write[k][0] else 0.0
        h, m = hit.get(k, [0, 0.0])[1], miss.get(k, [0, 0.0])[1]
        out[k] = dict(launches=n, fetch_bytes_raw=round(f_kb * 1024), fetch_bytes_corrected=round(2 * f_kb * 1024),
                      write_bytes_raw=round(w_kb * 1024), hbm_bytes_per_launch=round((2 * f_kb + w_kb) * 1024),
                      l2_hit_rate=round(h / (h + m), 4) if h + m else None,
                      l2_requests_per_launch=round(req[k][1] / req[k][0]) if k in req and req[k][0] else None)
        if k in dur:
            out[k]['avg_us'] = dur[k]
            out[k]['hbm_side_GBps'] = round(out[k]['hbm_bytes_per_launch'] / dur[k] / 1e3, 1)
            if mfma and k in mfma:
                out[k]['mfma_busy_cycles_per_launch'] = round(mfma[k][1] / mfma[k][0])
                out[k]['mfma_busy_frac'] = round(mfma[k][1] / mfma[k][0] / N_SIMD / (dur[k] * CLOCK_MHZ), 4)
    with open(dst + '.json', 'w') as f:
        import subprocess, datetime
        try:
            head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
        except OSError:
            head = None
        head = os.environ.get('S3D_HEAD', head)             # the GPU box has no .git: the caller passes the commit the tree was cut from
        json.dump(dict(head=head, date=datetime.date.today().isoformat(), command='bench.py --no-graphs --steps 4 --warmup 2 (cfg-2, batch 64), rocprofv3 --kernel-trace --pmc, '
                               'one pass per counter group', correction='fetch x2 (gfx950 wide-read tally), write raw',
                       kernels=out), f, indent=1)
    with open(dst + '.txt', 'w') as f:
        f.write('# HBM-side bytes per launch (FETCH_SIZE x2 corrected + WRITE_SIZE raw), L2 hit rate; cfg-2 step, eager launches\n')
        f.write('# avg_us: un-profiled duration in the replayed graph (rocprofv3 --stats); GB/s = total_MB / avg_us (HBM peak ~8000); '
                'MFMA% = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (avg_us x 2100 MHz)\n')
        f.write(f'{"launches":>8} {"fetch_MB":>9} {"write_MB":>9} {"total_MB":>9} {"L2hit":>6} {"avg_us":>8} {"GB/s":>7} {"MFMA%":>6}  kernel\n')
        nan = float('nan')
        for k, v in out.items():
            f.write(f'{v["launches"]:8d} {v["fetch_bytes_corrected"] / 1e6:9.2f} {v["write_bytes_raw"] / 1e6:9.2f} '
                    f'{v["hbm_bytes_per_launch"] / 1e6:9.2f} {v["l2_hit_rate"] if v["l2_hit_rate"] is not None else nan:6.3f} '
                    f'{v.get("avg_us", nan):8.2f} {v.get("hbm_side_GBps", nan):7.1f} {100 * v.get("mfma_busy_frac", nan):6.1f}  {k[:100]}\n')
    print(open(dst + '.txt').read())


if __name__ == '__main__':
    main()
