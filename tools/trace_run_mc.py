"""Workload for `rocprofv3 --kernel-trace`: PCG with IC(0) in multicolour order (colour sweeps), update by update (no graph):
    python tools/trace_run_mc.py [c2|c3|natural]       then tools/level_trace.py on the kernel-trace CSV."""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
if which == "c2":
    s = poisson.poisson_system(2, 256)
elif which == "natural":
    s = poisson.poisson_system(3, 100)
elif which == "nearly":                      # bipartite but for 0.5 % extra couplings (tools/mc_probe.py)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mc_probe_cases", str(ROOT / "tools" / "mc_probe.py"))
    sys.argv = [sys.argv[0], "none"]
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    s = mod._grid3d_with_extra_links(100, 0.005)
else:
    s = D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0))
s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
b = poisson.rhs(s.n, 0)
s.solve(b, max_iter=4, want_history=False)
s.solve(b, max_iter=24, want_history=False, flags=D._lib.NO_GRAPH)
torch.cuda.synchronize()
