#!/usr/bin/env python
"""Static check of a gfx9 assembly listing (hipcc -S --cuda-device-only): does any instruction READ (or overwrite) a VGPR
that an older vector-memory LOAD still has in flight, i.e. is an `s_waitcnt vmcnt(N)` missing or too weak?

Why (round 6): builds of the registration kernel in which the SLP vectoriser formed packed-f32 code were not
bitwise reproducible next to a second process on the GPU -- and only some of them (the same source with one more kernel
argument was clean).  A result that depends on memory latency is what a missing wait looks like: with the GPU to itself the
load has (always) landed by the time its register is read, next to a competitor it sometimes has not.  This tool models
the in-order `vmcnt` counter over straight-line code (state is dropped at labels and branches: a conservative miss, never
a false alarm from control flow) and the dword selection of packed instructions (`op_sel` / `op_sel_hi`: source i feeds
the low result from dword op_sel[i], the high result from dword op_sel_hi[i]).

    python tools/waitcnt_check.py kernel.s [function-name-substring] [--through-labels]
"""
import re
import sys

THROUGH = '--through-labels' in sys.argv   # keep the in-flight loads across labels / conditional branches (fall-through path)
REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
LOAD = re.compile(r'^(global|flat|buffer|scratch)_load|^(global|flat|buffer)_atomic.*\bsc0\b|^(global|flat|buffer)_atomic_\w+_rtn')
STORE = re.compile(r'^(global|flat|buffer|scratch)_store|^(global|flat|buffer)_atomic')
PACKED = re.compile(r'^v_pk_(fma|mul|add)_f32|^v_pk_mov_b32')


def regs_of(op):
    m = REG.search(op)
    if not m:
        return None
    if m.group(1) is not None:
        return [int(m.group(1))]
    return list(range(int(m.group(2)), int(m.group(3)) + 1))


def sel(mods, name, n, default):
    m = re.search(name + r':\[([01,]+)\]', mods)
    v = [int(x) for x in m.group(1).split(',')] if m else []
    return v + [default] * (n - len(v))


def check(path, want=None):
    text = open(path).read()
    funcs = re.split(r'\n(?=[_A-Za-z0-9$.]+:\s+; @)', text)
    total = 0
    for f in funcs:
        name = f.split(':', 1)[0].strip()
        if want and want not in name:
            continue
        if not re.search(r'\bs_endpgm\b', f):
            continue
        pending = []      # in issue order: ('load', set(dest regs), line no, text) | ('store', None, ..)
        lds = []          # LDS operations in issue order (they complete in order among themselves): (dest VGPRs, line, text);
                          # `lgkmcnt(N)` guarantees an LDS operation iff fewer than N + 1 LDS operations were issued at or
                          # after it (scalar loads share the counter and only make the guarantee stronger)
        viol, n_ins, n_wait = [], 0, 0
        for ln, line in enumerate(f.split('\n')):
            s = line.split(';')[0].strip()
            if not s or s.startswith('.') or s.startswith('//'):
                continue
            if s.endswith(':'):          # a label: another path may join here with other loads in flight
                if not THROUGH:
                    pending, lds = [], []
                continue
            op, _, rest = s.partition(' ')
            n_ins += 1
            if op.startswith('s_waitcnt'):
                n_wait += 1
                m = re.search(r'vmcnt\((\d+)\)', rest)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
                elif re.fullmatch(r'\s*(0x[0-9a-f]+|\d+)\s*', rest):   # raw immediate: treat as a full wait
                    pending, lds = [], []
                m = re.search(r'lgkmcnt\((\d+)\)', rest)
                if m:
                    keep = int(m.group(1))
                    lds = lds[len(lds) - keep:] if keep else []
                continue
            if op.startswith('s_cbranch') and THROUGH:
                continue                 # (--through-labels: follow the fall-through path with its loads in flight)
            if op.startswith('s_cbranch') or op.startswith('s_branch') or op in ('s_endpgm', 's_setpc_b64', 's_swappc_b64'):
                pending, lds = [], []
                continue
            parts = [p.strip() for p in re.split(r',(?![^\[]*\])', rest)]
            ops = [p for p in parts if p]
            mods = ' '.join(p for p in ops if ':' in p and not p.startswith('v[') and not p.startswith('s['))
            vops = [p for p in ops if re.match(r'^-?\|?v(\d+|\[)', p)]
            is_load, is_store = bool(LOAD.match(op)), bool(STORE.match(op)) and not LOAD.match(op)
            reads, writes = set(), set()
            if is_load:
                d = regs_of(vops[0]) if vops else None
                for p in vops[1:]:
                    reads.update(regs_of(p) or [])
                writes.update(d or [])
            elif is_store:
                for p in vops:
                    reads.update(regs_of(p) or [])
            elif op.startswith('ds_') or op.startswith('v_') or op.startswith('buffer_') or op.startswith('global_'):
                if PACKED.match(op) and len(vops) >= 2:
                    nsrc = len(vops) - 1
                    lo, hi = sel(mods, 'op_sel', nsrc, 0), sel(mods, 'op_sel_hi', nsrc, 1)
                    writes.update(regs_of(vops[0]) or [])
                    for i, p in enumerate(vops[1:]):
                        r = regs_of(p) or []
                        if len(r) == 2:
                            reads.update({r[lo[i]], r[hi[i]]})
                        else:
                            reads.update(r)
                else:
                    dst_first = not (op.startswith('ds_write') or op.startswith('ds_store') or op.startswith('v_cmp')
                                     or op.startswith('v_writelane') is False and False)
                    if op.startswith('ds_write') or op.startswith('ds_store'):
                        for p in vops:
                            reads.update(regs_of(p) or [])
                    else:
                        if vops:
                            writes.update(regs_of(vops[0]) or [])
                            # accumulate forms (v_fmac, v_mac, v_pk? handled) read their destination too
                            if re.match(r'^v_(fmac|mac|dot\w*c)_', op):
                                reads.update(regs_of(vops[0]) or [])
                        for p in vops[1:]:
                            reads.update(regs_of(p) or [])
            touched = reads | writes
            if touched:
                for kind, dest, pl, ptxt in pending:
                    if kind == 'load' and dest & touched:
                        older = sum(1 for k in pending[pending.index((kind, dest, pl, ptxt)):])
                        viol.append((ln, s, pl, ptxt, sorted(dest & touched), older))
                        break
                for dest, pl, ptxt in lds:
                    if dest & touched:
                        viol.append((ln, s, pl, ptxt, sorted(dest & touched), len(lds) - lds.index((dest, pl, ptxt))))
                        break
            if is_load:
                pending.append(('load', set(writes), ln, s))
            elif is_store:
                pending.append(('store', set(), ln, s))
            elif op.startswith('ds_'):
                has_dest = not (op.startswith('ds_write') or op.startswith('ds_store')) or '_rtn' in op
                lds.append((set(writes) if has_dest else set(), ln, s))
        print(f'{name[:70]}: {n_ins} instructions, {n_wait} s_waitcnt, {len(viol)} reads / overwrites of a register with its load still in flight')
        for ln, s, pl, ptxt, regs, older in viol[:12]:
            print(f'   line {ln}: {s}\n      touches v{regs} while in flight from line {pl}: {ptxt}   (needs vmcnt <= {older - 1})')
        total += len(viol)
    return total


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    sys.exit(1 if check(args[0], args[1] if len(args) > 1 else None) else 0)
