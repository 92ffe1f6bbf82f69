"""A/B probe for the K3 `done` hand-off: the stress loop of test_last_x_update_survives_multi_stream_contention,
counting mismatching solutions instead of asserting.  DPCG_LIBRARY=tools/libdpcg_r1.so runs the round-1 build."""
import sys

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd.batch import solve_batch
from oracle import oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
flags = D._lib.NO_SMALL | D._lib.NO_FUSE
mats = [O.poisson2d(96 + 2 * i) for i in range(32)]
systems = [D.CsrSystem.from_any(A) for A in mats]
for S in systems:
    S.set_preconditioner(D.Jacobi())
rhs = [torch.from_numpy(O.rhs(A.shape[0], i)).cuda() for i, A in enumerate(mats)]
single = [S.solve(b, flags=flags, want_history=False) for S, b in zip(systems, rhs)]
side = torch.cuda.Stream()
ga = torch.randn(4096, 4096, device="cuda")
gb = torch.randn(4096, 4096, device="cuda")
bad = 0
for rnd in range(rounds):
    with torch.cuda.stream(side):
        for _ in range(24):
            gb = torch.mm(ga, gb).mul_(1e-3)
    out = solve_batch(systems, rhs, flags=flags, n_streams=8)
    for r, s1 in zip(out, single):
        if not torch.equal(r.x, s1.x):
            bad += 1
            rows = int((r.x != s1.x).sum())
            print(f"round {rnd}: n={r.x.numel()} rows differing {rows}", flush=True)
side.synchronize()
print(f"{D._lib.LIB_PATH.name}: {bad} mismatching solutions in {rounds * len(systems)} solves")
