#!/usr/bin/env python3
"""Dev tool: timeline of the fused full-rank evaluation from a -DVB_FUSED_CLOCK build.

    VB_FR_FUSED=3 VB_FUSED_CLOCK_DUMP=/tmp/clk.bin python tools/fr_bench.py 1024 4096 gauss_full 20
    python tools/fused_clock.py /tmp/clk.bin

Per item: [start, end] in 100 MHz ticks (s_memrealtime), phase, block x | block z << 32.  Prints per phase the first
start, last end, mean / max lifetime, and a utilisation timeline (items in flight per phase every 10 us)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
t0 = a[:, 0].min()
start, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0      # us
phase = a[:, 2]
print('items %d, makespan %.1f us' % (len(a), end.max()))
for p in range(3):
    m = phase == p
    if not m.any():
        continue
    life = end[m] - start[m]
    print('phase %d: %4d items, first start %6.1f, last start %6.1f, first end %6.1f, last end %6.1f, lifetime mean %6.1f max %6.1f min %6.1f'
          % (p + 1, m.sum(), start[m].min(), start[m].max(), end[m].min(), end[m].max(), life.mean(), life.max(), life.min()))
step = 10.0
print('in flight per phase (every %.0f us):' % step)
for t in np.arange(0, end.max() + step, step):
    row = [int(np.sum((phase == p) & (start <= t) & (end > t))) for p in range(3)]
    print('  t=%6.1f  P1 %4d  P2 %4d  P3 %4d  total %4d' % (t, row[0], row[1], row[2], sum(row)))
if len(sys.argv) > 2:      # per-item dump of one phase
    p = int(sys.argv[2]) - 1
    for i in np.nonzero(phase == p)[0][:int(sys.argv[3]) if len(sys.argv) > 3 else 64]:
        print('  item %4d x %4d z %d: %.1f -> %.1f (%.1f us)' % (i, a[i, 3] & 0xffffffff, a[i, 3] >> 32, start[i], end[i], end[i] - start[i]))
