#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel count / avg / min / max duration (us).

rocprofv3 on this image writes rocpd .db files; this prints the same table `--stats` would and
is used to produce the summaries committed under profiles/.
"""
import sqlite3
import sys


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, s.arch_vgpr_count, "
        "s.sgpr_count, s.group_segment_size from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    stats = {}
    for name, st, en, gx, wx, vg, sg, lds in rows:
        stats.setdefault(name, []).append(((en - st) / 1e3, gx, wx, vg, sg, lds))
    total = sum(sum(x[0] for x in v[skip_first:]) for v in stats.values())
    print('%-78s %7s %9s %9s %9s %9s %6s %8s %5s %5s %6s' % (
        'kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'total_ms', 'pct', 'grid', 'wg', 'vgpr', 'lds'))
    for name, v in sorted(stats.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        vv = v[skip_first:] or v
        d = [x[0] for x in vv]
        short = name if len(name) <= 78 else name[:75] + '...'
        print('%-78s %7d %9.3f %9.3f %9.3f %9.3f %6.1f %8d %5d %5d %6d' % (
            short, len(d), sum(d) / len(d), min(d), max(d), sum(d) / 1e3,
            100 * sum(d) / max(total, 1e-9), vv[0][1], vv[0][2], vv[0][3], vv[0][5]))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
