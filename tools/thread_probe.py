"""Four host threads, each on its own stream, creating systems, attaching IC(0) / ICT / Jacobi and solving, over and over -- beside a
fifth that keeps calling torch.cuda.synchronize() and allocating: the device
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
stop_noise = threading.Event()
noise_refused = [0]


def noise():
    """A host thread of the application itself: device-wide waits and allocations while the library works on the other threads."""
    while not stop_noise.is_set():
        try:
            torch.cuda.synchronize()
            t = torch.empty(1 << 20, device="cuda")
            del t
            torch.cuda.synchronize()
        except Exception:                       # HIP refuses a device-wide wait while ANY stream captures: the application's own matter
            noise_refused[0] += 1


noisy = threading.Thread(target=noise)
noisy.start()
threads = [threading.Thread(target=run, args=(i, 12, out)) for i in range(4)]
[t.start() for t in threads]
[t.join() for t in threads]
stop_noise.set()
noisy.join()
bad = 0
for i in range(4):
    for a, p, it, x in out[i]:
        if it != alone[(a, p)][0] or not np.array_equal(x, alone[(a, p)][1]):
            bad += 1
            print("MISMATCH thread", i, "case", a, "precond", p, it, alone[(a, p)][0])
# phase 2: every thread solves its own batch (team / whole-solve launch forms), updates the values of its systems and solves again
from deeppreconditioning_amd.batch import solve_batch
import scipy.sparse as sp

bA = [O.poisson2d(120), O.poisson2d(48), O.poisson3d(20)]
bA2 = []
for q, A in enumerate(bA):
    d = np.random.default_rng(q).uniform(0.5, 2.0, A.shape[0])
    B = (sp.diags(d) @ A @ sp.diags(d)).tocsr()
    B.sort_indices()
    bA2.append(B)
brhs = [O.rhs(A.shape[0], 7 + q) for q, A in enumerate(bA)]


def batch_run(i, rounds, out):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        S = [D.CsrSystem.from_any(A, reorder=None) for A in bA]
        b = [torch.from_numpy(v).cuda() for v in brhs]
        res = []
        for r in range(rounds):
            mats = bA if r % 2 == 0 else bA2
            for s_, A in zip(S, mats):
                s_.update_values(A.data)
                s_.set_preconditioner(D.Jacobi())
            res.append([(x.iterations, x.x.cpu().numpy()) for x in solve_batch(S, b)])
        for s_ in S:
            s_.close()
        stream.synchronize()
    out[i] = res


ref = {}
for parity, mats in ((0, bA), (1, bA2)):
    S = [D.CsrSystem.from_any(A, reorder=None) for A in mats]
    for s_ in S:
        s_.set_preconditioner(D.Jacobi())
    ref[parity] = [(x.iterations, x.x.cpu().numpy()) for x in solve_batch(S, [torch.from_numpy(v).cuda() for v in brhs])]
    for s_ in S:
        s_.close()
out2 = {}
threads = [threading.Thread(target=batch_run, args=(i, 6, out2)) for i in range(4)]
[t.start() for t in threads]
[t.join() for t in threads]
n2 = 0
for i in range(4):
    for r, res in enumerate(out2[i]):
        for (it, x), (it0, x0) in zip(res, ref[r % 2]):
            n2 += 1
            if it != it0 or not np.array_equal(x, x0):
                bad += 1
                print("BATCH MISMATCH thread", i, "round", r, it, it0)
print(f"thread_probe: {sum(len(v) for v in out.values())} solves on 4 threads, {bad} mismatches ({n2} more in concurrent batches with update_values)")
raise SystemExit(1 if bad else 0)
