#!/usr/bin/env python
"""Per-kernel digest of the PMC passes of tools/pmc_cfg3.sh:  python tools/pmc_cfg3_summary.py gpurun_out/pmc3b"""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    agg = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(path)):
        name = re.sub(r'\(S3d.*$', '', row['Kernel_Name']).replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        agg[name][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] in ('SQ_WAVES',):
            agg[name]['calls'] += 1
    return agg


src = sys.argv[1]
d = {}
for p in ['sq1', 'sq2', 'sq3', 'tcc']:
    for k, v in load(f'{src}/{p}/p_counter_collection.csv').items():
        d.setdefault(k, {}).update(v)
print('kernel | wave-cycles share | MFMA busy / SIMD-cycle | wait_any | wait_inst | VALU/MFMA | LDS/MFMA | VMEM/MFMA | LDS conflict | L2 hit')
tot = sum(v.get('SQ_WAVE_CYCLES', 0) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:22]:
    wc = max(v.get('SQ_WAVE_CYCLES', 1), 1)
    mf = max(v.get('SQ_INSTS_MFMA', 0), 1)
    # SQ_BUSY_CYCLES is summed over the XCD shader engines (32 per chip); SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(v.get('GRBM_GUI_ACTIVE', 1) / 8, 1)
    print(f"{k[:64]:64s} busy_cyc {v.get('SQ_BUSY_CYCLES', 0):.3e}  mfma_busy {busy:5.2f}  wait_any {v.get('SQ_WAIT_ANY', 0) / wc:4.2f}  "
          f"wait_inst {v.get('SQ_WAIT_INST_ANY', 0) / wc:4.2f}  valu/mfma {v.get('SQ_INSTS_VALU', 0) / mf:6.2f}  lds/mfma {v.get('SQ_INSTS_LDS', 0) / mf:5.2f}  "
          f"vmem/mfma {v.get('SQ_INSTS_VMEM_RD', 0) / mf:5.2f}  conflict {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):4.2f}  "
          f"L2hit {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):5.3f}")
