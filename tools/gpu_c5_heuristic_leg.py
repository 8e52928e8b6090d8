import sys, json
sys.path.insert(0, ".")
import bench
r = bench.c5_heuristic_leg()
print(json.dumps(r, indent=1)[:1800])
