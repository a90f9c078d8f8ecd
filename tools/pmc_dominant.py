#!/usr/bin/env python
"""PMC evidence of the dominant conv kernel from separate rocprofv3 --pmc passes (rocpd SQLite results).
    python tools/pmc_dominant.py <kernel-name-substring> <workload label> pass1.db pass2.db ... > profiles/r02_dominant_pmc.json
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / dispatches (gfx950: FETCH_SIZE reads 1/2 of wide
coalesced reads, MI355X_MICROARCH.md); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x active cycles per
XCD), active cycles per XCD = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import json
import sqlite3
import sys


def main():
    pat, workload, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    c, disp, dur = {}, {}, []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for name, ctr, tot, n in cur.execute(
                'select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection '
                'where kernel_name like ? group by kernel_name, counter_name', (f'%{pat}%',)):
            c[ctr] = c.get(ctr, 0.0) + tot
            disp[ctr] = disp.get(ctr, 0) + n
        for avg, n in cur.execute('select avg(duration), count(*) from kernels where name like ?', (f'%{pat}%',)):
            if n:
                dur.append(avg)
    per = {k: c[k] / max(1, disp[k]) for k in c}
    out = {'kernel': pat, 'workload': workload, 'dispatches_per_pass': disp,
           'command': 'rocprofv3 --kernel-trace --pmc <set> -- python bench.py --streams 1 --steps 2 --warmup 1 --no-parity '
                      '[--pairs-per-step as the workload label says] (one pass per counter set, tools/evidence.sh)',
           'per_dispatch': per, 'avg_duration_us_profiled': sum(dur) / max(1, len(dur)) / 1e3}
    if 'FETCH_SIZE' in per and 'WRITE_SIZE' in per:
        out['hbm_bytes_per_launch'] = (2.0 * per['FETCH_SIZE'] + per['WRITE_SIZE']) * 1024.0
        out['hbm_note'] = ('2 x FETCH_SIZE (gfx950 correction; exact on a 2-GB streaming read, profiles/r05_mall_probe_counters.txt) + '
                           'WRITE_SIZE, KB -> bytes; both count what crosses L2 <-> fabric, Infinity-Cache hits included')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in per and 'GRBM_GUI_ACTIVE' in per:
        cyc = per['GRBM_GUI_ACTIVE'] / 8.0
        out['active_cycles_per_xcd'] = cyc
        out['mfma_busy'] = per['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)
        if out['avg_duration_us_profiled']:
            out['shader_clock_ghz'] = cyc / (out['avg_duration_us_profiled'] * 1e3)
    if 'SQ_WAVE_CYCLES' in per:
        for k in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY'):
            if k in per:
                out[k.lower() + '_frac_of_wave_cycles'] = per[k] / per['SQ_WAVE_CYCLES']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
