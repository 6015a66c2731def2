"""Per-node-sweep cost of k_dw_solve on domains whose tables do / do not fit one XCD's L2 (4 MB):
LowerColorado (408 mainstem nodes x 32 KB of tables = 13 MB) against the small goldens (<= 0.5 MB).
Node-sweep and depth-function counts come from the host instantiation (tools only; DW_ORACLE_COUNTERS)."""
import ctypes as C, os, re, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--count":
    import test_diffusive as T
    name = sys.argv[2]
    ins = T.load_lowercolorado(12)[0] if name == "lowercolorado" else (T.load_lowercolorado(12, "diffusive_lowercolorado_nat.npz")[0] if name == "lowercolorado_nat" else T.load_small(name)[0])
    T.call_c(T.host_oracle(), "dw_oracle_diffnw", ins)
    sys.exit(0)
import test_diffusive as T
from troute_amd.routing.fast_reach import diffusive as D
for name in ("chain1", "y3", "comb", "comb_nat", "lowercolorado", "lowercolorado_nat"):
    ins = T.load_lowercolorado(12)[0] if name == "lowercolorado" else (T.load_lowercolorado(12, "diffusive_lowercolorado_nat.npz")[0] if name == "lowercolorado_nat" else T.load_small(name)[0])
    D.compute_diffusive(ins)
    D.compute_diffusive(ins)
    tables_ms, solve_ms = D.last_timing()
    err = subprocess.run([sys.executable, __file__, "--count", name], env=dict(os.environ, DW_ORACLE_COUNTERS="1"),
                         capture_output=True, text=True).stderr
    m = re.search(r"sub-steps (\d+) node sweeps (\d+) funcd (\d+)", err)
    ss, ns, fd = (int(x) for x in m.groups())
    print(f"{name:18s} solve {solve_ms:9.2f} ms  sub-steps {ss:6d}  node sweeps {ns:8d}  funcd {fd:8d} ({fd/ns:.2f}/sweep)  "
          f"{solve_ms*1e3/ns:6.2f} us per node sweep")
