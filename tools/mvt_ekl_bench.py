#!/usr/bin/env python3
"""Dev tool: MultivariateT(256) + ExclusiveKL at N = 16 384 in throughput mode (bench.py's mvt_ekl leg by itself)."""
import sys
sys.path.insert(0, '.')
import bench
import viabel_amd as vb
r = bench.mvt_ekl_leg(vb, calls=40)
print('MultivariateT + ExclusiveKL, D=256 N=16384: throughput mode %.3f ms per call, reference-identical mode %.2f ms'
      % (r['throughput_mode']['ms_per_call'], r['parity_mode']['ms_per_call']))
