import sys, json
sys.path.insert(0, '.')
import bench
import viabel_amd as vb
from viabel_amd import _lib
eng = _lib.default_engine()
for i in range(2):
    out = bench.api_call_leg(eng, vb, calls=100)
    print(i, {k: round(v, 1) for k, v in out.items() if k.endswith('_us_per_call')}, {k: round(v, 1) for k, v in out['split'].items()})
