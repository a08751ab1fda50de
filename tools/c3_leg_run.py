#!/usr/bin/env python3
"""Dev tool: bench.py's c3_mvt_dis leg alone (throughput and reference-identical modes)."""
import sys
sys.path.insert(0, '.')
import bench
import viabel_amd as vb
r = bench.c3_leg(vb)
print({k: round(v['ms_per_call'], 3) for k, v in r.items() if isinstance(v, dict) and 'ms_per_call' in v})
