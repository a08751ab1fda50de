import json, sys
sys.path.insert(0, '.')
import bench
import viabel_amd as vb
print(json.dumps(bench.source_model_leg(vb), indent=0)[:1600])
