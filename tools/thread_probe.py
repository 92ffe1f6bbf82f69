"""Four host threads, each on its own stream, creating systems, attaching IC(0) / ICT / Jacobi and solving, over and over: the device
block cache (thread-local scopes, one process-wide pool) and the setup routines under concurrency.  Every result must equal the
one computed alone."""
import threading
import numpy as np
import torch
import deeppreconditioning_amd as D
from oracle import oracle as O

cases = [O.poisson2d(96), O.unstructured_like(O.poisson3d(24), 1), O.poisson3d(40), O.poisson2d(300)]
rhs = [O.rhs(A.shape[0], i) for i, A in enumerate(cases)]
pcs = [lambda: D.IC0("solve"), lambda: D.ICT("solve"), lambda: D.Jacobi(), lambda: D.IC0("multiply")]


def run(i, rounds, out):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        res = []
        for r in range(rounds):
            A, b = cases[(i + r) % 4], torch.from_numpy(rhs[(i + r) % 4]).cuda()
            S = D.CsrSystem.from_any(A)
            S.set_preconditioner(pcs[(i + 2 * r) % 4]())
            x = S.solve(b, max_iter=200)
            res.append(((i + r) % 4, (i + 2 * r) % 4, x.iterations, x.x.cpu().numpy()))
            S.close()
        stream.synchronize()
    out[i] = res


alone = {}
for a in range(4):
    for p in range(4):
        S = D.CsrSystem.from_any(cases[a])
        S.set_preconditioner(pcs[p]())
        x = S.solve(torch.from_numpy(rhs[a]).cuda(), max_iter=200)
        alone[(a, p)] = (x.iterations, x.x.cpu().numpy())
        S.close()
out = {}
threads = [threading.Thread(target=run, args=(i, 12, out)) for i in range(4)]
[t.start() for t in threads]
[t.join() for t in threads]
bad = 0
for i in range(4):
    for a, p, it, x in out[i]:
        if it != alone[(a, p)][0] or not np.array_equal(x, alone[(a, p)][1]):
            bad += 1
            print("MISMATCH thread", i, "case", a, "precond", p, it, alone[(a, p)][0])
print(f"thread_probe: {sum(len(v) for v in out.values())} solves on 4 threads, {bad} mismatches")
raise SystemExit(1 if bad else 0)
