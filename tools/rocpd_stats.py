#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per (kernel, grid) count / avg / min / max duration.

rocprofv3 on this image writes rocpd .db files; this prints the table `--stats` would (split by
launch geometry, because the bench launches the same kernel for 16-evaluation batches and for
single blocking calls) and is used to produce the summaries committed under profiles/.
With a counter name as 2nd argument it also prints the per-launch average of that PMC counter.
"""
import sqlite3
import sys


def main(path, counter=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, "
        "s.arch_vgpr_count, s.sgpr_count, s.group_segment_size, d.event_id from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    pmc = {}
    if counter:
        for ev, val in cur.execute(
                "select p.event_id, p.value from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id "
                "where i.name = ?", (counter,)):
            pmc[ev] = pmc.get(ev, 0.0) + val
    stats = {}
    for name, st, en, gx, gy, wx, vg, sg, lds, ev in rows:
        stats.setdefault((name, gx, gy), []).append(((en - st) / 1e3, wx, vg, lds, pmc.get(ev)))
    total = sum(sum(x[0] for x in v) for v in stats.values())
    hdr = '%-64s %9s %7s %9s %9s %9s %9s %6s %5s %5s %6s' % (
        'kernel', 'grid(x,y)', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', 'pct', 'wg', 'vgpr', 'lds')
    if counter:
        hdr += ' %14s' % ('avg_' + counter)
    print(hdr)
    for (name, gx, gy), v in sorted(stats.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        d = [x[0] for x in v]
        short = name if len(name) <= 64 else name[:61] + '...'
        line = '%-64s %9s %7d %9.3f %9.3f %9.3f %9.3f %6.1f %5d %5d %6d' % (
            short, '%dx%d' % (gx, gy), len(d), sum(d) / len(d), min(d), max(d), sum(d) / 1e3,
            100 * sum(d) / max(total, 1e-9), v[0][1], v[0][2], v[0][3])
        if counter:
            vals = [x[4] for x in v if x[4] is not None]
            line += ' %14.1f' % (sum(vals) / len(vals)) if vals else ' %14s' % '-'
        print(line)


def timeline(path, first, count):
    """Dispatches `first` .. `first + count` in start order: offset from the first one's start, duration, and the gap
    to the previous dispatch's end (negative: they overlapped -- another stream)."""
    db = sqlite3.connect(path)
    rows = db.cursor().execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x, d.queue_id from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[first:first + count]
    t0, prev_end = rows[0][1], rows[0][1]
    print('%9s %9s %8s %6s  %s' % ('start_us', 'dur_us', 'gap_us', 'queue', 'kernel (grid x)'))
    for name, st, en, gx, q in rows:
        short = name[:70]
        print('%9.1f %9.1f %8.1f %6s  %s (%d)' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, q, short, gx))
        prev_end = max(prev_end, en)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == '--timeline':
        timeline(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]))
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
