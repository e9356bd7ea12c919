"""How long does the stand-in collective (s3d_debug_paced_copy) really take?  (tools only; MI355X)"""
import ctypes, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple3d_former_amd import _lib as L
lib = L.lib()
src = torch.zeros(32 << 20, dtype=torch.float32, device='cuda'); dst = torch.empty_like(src)
for mb in (7.5, 21, 28, 85.6):
    for gbps in (100.0, 286.0, 600.0, 2000.0):
        n = int(mb * 1e6) // 16 * 16
        def run():
            L.check(lib.s3d_debug_paced_copy(L.ptr(dst), L.ptr(src), ctypes.c_long(n), ctypes.c_float(gbps), L.current_stream()), 'paced')
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f'{mb:6.1f} MB at {gbps:6.0f} GB/s requested: {us:7.1f} us = {n / us / 1e3:7.1f} GB/s achieved ({min(64, max(4, int(gbps / 18) + 1))} workgroups)')
