"""Register / LDS / scratch usage of the kernels in libs3d_hip.so (llvm-readelf --notes on the embedded gfx950 code objects).
    python tools/kernel_resources.py [name substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_host_cpu import _gfx950_code_objects
import simple3d_former_amd._lib as L

def main():
    pats = sys.argv[1:]
    objs = _gfx950_code_objects(os.environ.get('S3D_LIB_PATH', L.LIB_PATH))
    rows = []
    for i, o in enumerate(objs):
        with tempfile.NamedTemporaryFile(suffix='.elf', delete=False) as f:
            f.write(o)
        notes = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True, check=True).stdout
        os.unlink(f.name)
        cur = {}
        for ln in notes.splitlines():
            m = re.match(r'\s*-?\s*\.(\w+):\s+(\S+)', ln)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == 'name' and v.startswith('_Z'):
                cur['name'] = v
            elif k in ('vgpr_count', 'agpr_count', 'sgpr_count', 'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size'):
                cur[k] = int(v)
            if k == 'vgpr_count' and 'name' in cur:
                pass
            if k == 'wavefront_size' and 'name' in cur:
                rows.append(cur); cur = {}
    for r in rows:
        dem = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in dem for p in pats):
            continue
        print(f"vgpr {r.get('vgpr_count', -1):4d} agpr {r.get('agpr_count', -1):4d} sgpr {r.get('sgpr_count', -1):4d} scratch {r.get('private_segment_fixed_size', -1):4d} "
              f"lds {r.get('group_segment_fixed_size', -1):6d} wg {r.get('max_flat_workgroup_size', -1):4d}  {dem[:150]}")

if __name__ == '__main__':
    main()
