import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for passes in (2, 3, 3):
    r = bench.product_loop_case(4, 8, 32, True, 'batch', True, device=0, reps=1, passes=passes)
    print(os.environ.get('TAG'), passes, {k: round(r[k], 3) for k in ('wall_ms_per_snapshot', 'wall_ms_per_snapshot_resident', 'resident_over_kernel_only_wall', 'host_ms_per_snapshot')}, flush=True)
