"""Static check of the un-waited LDS transpose reads (attn_frag.h gather_issue_2x2 / gather_wait): in the device assembly, no instruction between
a ds_read_b64_tr_b16 and the next `s_waitcnt lgkmcnt(0)` may READ or WRITE that read's destination registers (hipcc does not know the
registers are still in flight: a copy it places there reads stale data whenever the LDS is slower than the copy).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DS3D_EXPERIMENTAL_TILES -S --cuda-device-only -o /tmp/attn.s attention.hip
    python tools/r6/asm_async_read_check.py /tmp/attn.s"""
import re, sys

def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()

def main(path):
    kernel, pending, bad, total = None, {}, [], 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r'^(_Z\w+):', s)
        if m: kernel, pending = m.group(1), {}; continue
        if not s or s.startswith(('.', ';', '//')): continue
        s = s.split(';')[0].split('//')[0].strip()
        op, _, rest = s.partition(' ')
        toks = [t.strip() for t in re.split(r'[,\s]+', rest) if t.strip()]
        if op == 's_waitcnt' and 'lgkmcnt(0)' in rest: pending = {}; continue
        if op in ('s_barrier', 's_endpgm', 's_branch') or op.startswith('s_cbranch'):
            if pending and op != 's_barrier': bad.append((kernel, ln, s, 'branch with reads in flight'))
            continue
        touched = set()
        for t in toks: touched |= regs(t)
        hit = [r for r in touched if r in pending]
        if hit: bad.append((kernel, ln, s, f'touches v{sorted(hit)} in flight since line {pending[hit[0]]}'))
        if op == 'ds_read_b64_tr_b16':
            total += 1
            for r in regs(toks[0]): pending[r] = ln
    print(f'{total} ds_read_b64_tr_b16, {len(bad)} hazards')
    seen = {}
    for k, ln, s, why in bad:
        seen.setdefault(k, []).append((ln, s, why))
    for k, v in seen.items():
        print(f'  {k[:110]}: {len(v)}')
        for ln, s, why in v[:6]: print(f'      {ln}: {s}    <- {why}')
    return 1 if bad else 0

if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
