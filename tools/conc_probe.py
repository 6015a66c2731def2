"""Probe: do step launches of two independent plans on two streams overlap on one GPU?"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic, sharding
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 16
part = sharding.partition(to, world)
rs = []
for rank in (1, 2, 3, 4):
    r = ShardedRouter(to, params, rank=rank, world=world, partition=part)
    r.upload(288, qlat, q0)
    rs.append(r)
def seq(k):
    t0 = time.perf_counter()
    for r in rs[:k]:
        r.plan0.route_device(288, 12, True)
    return (time.perf_counter() - t0) * 1e3
def conc(k):
    t0 = time.perf_counter()
    for r in rs[:k]:
        r.plan0.route_begin(288, 12, True)
    for t in range(1, 289):
        for r in rs[:k]:
            r.plan0.route_advance(t)
    for r in rs[:k]:
        r.plan0.route_end()
    return (time.perf_counter() - t0) * 1e3
for k in (1, 2, 4):
    seq(k); conc(k)
    print(f"world {world} rows/plan {rs[0].rows0.size}: {k} plans sequential {seq(k):.2f} ms, interleaved on {k} streams {conc(k):.2f} ms")
