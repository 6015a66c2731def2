"""Probe: phase-0 time of one rank of an 8-way partition, with and without a second plan alive/active."""
import sys, os, numpy as np
if len(sys.argv) > 2:
    import torch  # before libtrmc: torch bundles its own HIP runtime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from troute_amd import synthetic, sharding
from troute_amd.distributed import ShardedRouter
net = synthetic.generate(cache_dir="/tmp/trmc_cache")
to, params, qlat = net["to"], net["params"], net["qlat"]
nseg = to.shape[0]
q0 = np.zeros((nseg, 3), np.float32)
part = sharding.partition(to, 8)
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
r = ShardedRouter(to, params, rank=rank, world=8, partition=part)
r.upload(288, qlat, q0)
for _ in range(3):
    st = r.plan0.route_device(288, 12, True)
print("rank", rank, "plan0 alone ms_main", round(st["ms_main"], 2), "plan1 exists:", r.plan1 is not None)
if r.plan1 is not None:
    bf = np.zeros((int(r.boundary1.sum()), 288, 3), np.float32)
    r.plan1.upload_forcing(288, qlat[r.rows1], q0[r.rows1], bf)
    st1 = r.plan1.route_device(288, 12, True)
    st1 = r.plan1.route_device(288, 12, True)
    print("  plan1 alone ms_main", round(st1["ms_main"], 2))
    st = r.plan0.route_device(288, 12, True)
    print("  plan0 after plan1 ran: ms_main", round(st["ms_main"], 2))
    # both at once through the async API
    for _ in range(2):
        r.plan0.route_begin(288, 12, True); r.plan1.route_begin(288, 12, True)
        r.plan0.route_advance(288); r.plan1.route_advance(288)
        a = r.plan0.route_end(); b = r.plan1.route_end()
    print("  concurrent: plan0", round(a["ms_main"], 2), "plan1", round(b["ms_main"], 2))
    for _ in range(2):
        r.plan0.route_begin(288, 12, True); r.plan1.route_begin(288, 12, True)
        for t in range(36, 289, 36):
            r.plan0.route_advance(t); r.plan1.route_advance(t)
        a = r.plan0.route_end(); b = r.plan1.route_end()
    print("  concurrent chunked: plan0", round(a["ms_main"], 2), "plan1", round(b["ms_main"], 2))
r.close()
if len(sys.argv) > 2:
    import torch
    r = ShardedRouter(to, params, rank=rank, world=8, partition=part)
    r.upload(288, qlat, q0)
    bf = np.zeros((int(r.boundary1.sum()), 288, 3), np.float32)
    r.plan1.upload_forcing(288, qlat[r.rows1], q0[r.rows1], bf)
    dev = torch.device("cuda", 0)
    s0 = torch.cuda.ExternalStream(r.plan0.stream(), device=dev)
    s1 = torch.cuda.ExternalStream(r.plan1.stream(), device=dev)
    sc = torch.cuda.Stream(device=dev)
    buf = torch.zeros((1000, 36), device=dev); buf2 = torch.zeros((8, 1000, 36), device=dev)
    def run(mode):
        for _ in range(2):
            r.plan0.route_begin(288, 12, True); r.plan1.route_begin(288, 12, True)
            for t in range(36, 289, 36):
                r.plan0.route_advance(t)
                if "e" in mode:        # s0 -> s1 dependency through a torch event
                    ev = torch.cuda.Event(); ev.record(s0); s1.wait_event(ev)
                if "c" in mode:        # s0 -> sc -> s1 with a copy on sc
                    ev = torch.cuda.Event(); ev.record(s0); sc.wait_event(ev)
                    with torch.cuda.stream(sc):
                        buf2[3].copy_(buf)
                    ev2 = torch.cuda.Event(); ev2.record(sc); s1.wait_event(ev2)
                if "k" in mode:        # a torch kernel on s1
                    with torch.cuda.stream(s1):
                        buf.add_(1.0)
                r.plan1.route_advance(t)
            a = r.plan0.route_end(); b = r.plan1.route_end()
        print(f"  torch mode '{mode}': plan0", round(a["ms_main"], 2), "plan1", round(b["ms_main"], 2))
    for mode in ("", "e", "c", "k", "ck"):
        run(mode)
    r.close()
if len(sys.argv) > 2:
    import time
    r = ShardedRouter(to, params, rank=rank, world=8, partition=part)
    r.enable_device_exchange(torch, dev)
    r.upload(288, qlat, q0)
    r.upload_trunk()
    def ag(out, t):
        out[rank].copy_(t)
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.route_on_device(12, True, ag)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("  route_on_device trivial all_gather:", round(el * 1e3, 2), "ms; plan0", round(r.last_stats["phase0"]["ms_main"], 2), "plan1", round(r.last_stats["phase1"]["ms_main"], 2))
    def ag2(out, t):
        out.zero_()
        out[rank].copy_(t)
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.route_on_device(12, True, ag2)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("  route_on_device zero+copy all_gather:", round(el * 1e3, 2), "ms; plan0", round(r.last_stats["phase0"]["ms_main"], 2), "plan1", round(r.last_stats["phase1"]["ms_main"], 2))
    r.close()
if len(sys.argv) > 2:
    r = ShardedRouter(to, params, rank=rank, world=8, partition=part)
    r.enable_device_exchange(torch, dev)
    r.upload(288, qlat, q0)
    r.upload_trunk()
    for val in (0.5, 50.0):
        def ag3(out, t, val=val):
            out.fill_(val)
            out[rank].copy_(t)
        for k in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.route_on_device(12, True, ag3)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"  route_on_device inflow {val}:", round(el * 1e3, 2), "ms; plan0", round(r.last_stats["phase0"]["ms_main"], 2), "plan1", round(r.last_stats["phase1"]["ms_main"], 2))
        it = r.plan1.download_iterations()
        print("     trunk iterations hist", np.bincount(it, minlength=8)[:8], "max", it.max())
    r.close()
