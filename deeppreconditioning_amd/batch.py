"""Batches of independent pressure systems: one GPU (interleaved streams) and one process per GPU.

The reference loops over samples one by one (train.py:90-108, test.py:121-149); systems never
couple, so the batch shards with no data-path collective: system s -> rank s mod world.  RCCL (the
"nccl" backend of torch.distributed on ROCm) is used only to scatter the batch description and to
gather the fixed-size result records.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L
from .operators import CsrSystem, SolveResult, _dev_ptr


def solve_batch(systems: list[CsrSystem], rhs: list[torch.Tensor], x0: list | None = None, *, rtol_sq: float = 1e-8,
                atol_sq: float = 0.0, max_iter: int = 1024, flags: int = 0, n_streams: int = 4) -> list[SolveResult]:
    """Solve independent systems on ONE GPU, interleaved over `n_streams` HIP streams
    (dpcg_solve_batch): small systems overlap each other's launch and reduction latency."""
    count = len(systems)
    if count == 0:
        return []
    if len(rhs) != count:
        raise ValueError("one right-hand side per system")
    dev = systems[0].device

    def as_vec(system, v):      # fast path: already an fp64 vector on the device (the common case in a batch)
        if (isinstance(v, torch.Tensor) and v.dtype == torch.float64 and v.device == system.device and v.dim() == 1
                and v.numel() == system.n and v.is_contiguous()):
            return v
        return system._vec(v)

    bs = [as_vec(s, b) for s, b in zip(systems, rhs)]
    x0s = None if x0 is None else [None if v is None else as_vec(s, v) for s, v in zip(systems, x0)]
    # one allocation for all solutions; the results are views of it (the library only copies into x: 8-B alignment)
    sizes = [s.n for s in systems]
    flat = torch.empty(sum(sizes), dtype=torch.float64, device=dev)
    xs = flat.split(sizes)
    PA = C.c_void_p * count
    handles = PA(*[s._h.value for s in systems])
    b_arr = PA(*[b.data_ptr() for b in bs])
    base, x_ptrs, off = flat.data_ptr(), [], 0
    for n in sizes:
        x_ptrs.append(base + 8 * off)
        off += n
    x_arr = PA(*x_ptrs)
    x0_arr = None if x0s is None else PA(*[(0 if v is None else v.data_ptr()) for v in x0s])
    iters = (C.c_int * count)()
    status = (C.c_int * count)()
    res = (C.c_double * count)()
    sec = (C.c_double * count)()
    # inputs were produced on torch's stream; the batch runs on its own.  (The STREAM is waited for, not the device: a device-wide
    # wait from this host thread would void a graph capture in progress on another one.)
    torch.cuda.current_stream(dev).synchronize()
    with torch.cuda.device(dev):
        L.check(L.lib().dpcg_solve_batch(count, handles, b_arr, x0_arr, x_arr, rtol_sq, atol_sq, int(max_iter),
                                         int(flags), int(n_streams), iters, res, sec, status))
    no_history = np.empty(0)
    return [SolveResult(xs[i], iters[i], status[i], res[i], sec[i], no_history) for i in range(count)]


@dataclass
class SystemSpec:
    """What rank 0 scatters for a synthetic system: it is rebuilt in the owner's HBM, so a 256^3
    system (1.4 GB of CSR) costs 24 bytes on the wire instead of crossing xGMI."""
    dim: int
    n: int
    seed: int


def shard(count: int, rank: int, world: int) -> list[int]:
    """Indices of the systems rank `rank` owns: s mod world == rank."""
    return list(range(rank, count, world))


def solve_specs_local(specs: list[SystemSpec], *, precond: str = "jacobi", rtol_sq: float = 1e-8, max_iter: int = 1024,
                      flags: int = 0, reuse_matrix: bool = True, concurrent: int = 4) -> np.ndarray:
    """Solve this rank's systems.  Returns records (len(specs), 4): [iterations, status, final_res, seconds].

    Systems of one (dim, n) share ONE generated matrix in HBM (the handles borrow the same CSR arrays; only their work
    vectors differ), and up to `concurrent` of them are in flight at once on separate streams (`solve_batch`): another
    system's kernels fill the launch boundaries and ramps of this one's (measured: 4 x 1M-DoF +21 %, 8 x 256^2 2.5x
    aggregate iterations/s).  `concurrent=1` solves one after another."""
    from . import poisson
    from .operators import IC0, Jacobi
    out = np.zeros((len(specs), 4), dtype=np.float64)
    groups: dict[tuple[int, int], list[int]] = {}
    for i, sp in enumerate(specs):
        groups.setdefault((sp.dim, sp.n) if reuse_matrix else (sp.dim, sp.n, i), []).append(i)
    for key, members in groups.items():
        dim, n = key[0], key[1]
        rowptr, col, val = poisson.poisson_csr(dim, n)
        width = max(1, min(concurrent, len(members)))
        handles = [CsrSystem(rowptr, col, val, rowptr.numel() - 1) for _ in range(width)]    # borrowed arrays
        for h in handles:
            h.set_preconditioner(Jacobi() if precond == "jacobi" else (IC0("solve") if precond == "ic0" else None))
        for start in range(0, len(members), width):
            chunk = members[start:start + width]
            rhs = [poisson.rhs(handles[0].n, specs[i].seed, handles[0].device) for i in chunk]
            if len(chunk) == 1:
                results = [handles[0].solve(rhs[0], rtol_sq=rtol_sq, max_iter=max_iter, flags=flags, want_history=False)]
            else:
                results = solve_batch(handles[:len(chunk)], rhs, rtol_sq=rtol_sq, max_iter=max_iter, flags=flags,
                                      n_streams=len(chunk))
            for i, r in zip(chunk, results):
                out[i] = (r.iterations, r.status, r.final_res, r.seconds)
        for h in handles:
            h.close()
        del handles, rowptr, col, val
    return out


def _comm_device():
    """Where collective payloads live: HBM for the nccl (= RCCL) backend, host memory for gloo."""
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def scatter_specs(specs: list[SystemSpec] | None) -> tuple[list[SystemSpec], list[int], int]:
    """The scatter of a synthetic batch: all ranks call it, rank 0 passes the batch, the others None.
    The spec table (3 int64 per system) is broadcast; returns (this rank's specs, their batch indices, batch size)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _comm_device()
    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        meta[0] = len(specs)
    dist.broadcast(meta, 0)
    count = int(meta.item())
    table = torch.zeros((count, 3), dtype=torch.int64, device=dev)
    if rank == 0:
        table.copy_(torch.tensor([[s.dim, s.n, s.seed] for s in specs], dtype=torch.int64).reshape(count, 3))
    dist.broadcast(table, 0)
    mine = shard(count, rank, world)
    rows = table.cpu().numpy()
    return [SystemSpec(*map(int, rows[i])) for i in mine], mine, count


def gather_records(local: np.ndarray, count: int, width: int = 4) -> np.ndarray:
    """The gather: every rank contributes the fixed-size records of the systems it owns (in the order
    of `shard`), `all_gather` over the backend, every rank returns the (count, width) table in batch order."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _comm_device()
    mine = shard(count, rank, world)
    per_rank = max((count + world - 1) // world, 1)
    buf = torch.full((per_rank, width), -1.0, dtype=torch.float64, device=dev)
    if len(mine):
        buf[: len(mine)] = torch.from_numpy(np.asarray(local, dtype=np.float64).reshape(len(mine), width)).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    out = np.zeros((count, width), dtype=np.float64)
    for r in range(world):
        idx = shard(count, r, world)
        out[idx] = gathered[r].cpu().numpy()[: len(idx)]
    return out


def solve_specs_distributed(specs: list[SystemSpec] | None, local_solver=None, **kw) -> np.ndarray | None:
    """All ranks call this; rank 0 passes the batch, the others None.

    scatter: the spec table is broadcast (3 int64 per system; `scatter_specs`).  Solve: rank r takes systems
    r, r+world, ... with no communication.  gather: fixed-size result records are all-gathered (`gather_records`);
    every rank returns the full (count, 4) table in batch order.  `local_solver(specs, **kw)`
    defaults to `solve_specs_local` (tests inject a stand-in to exercise the sharding on CPU/gloo).
    """
    my_specs, _, count = scatter_specs(specs)
    local = (local_solver or solve_specs_local)(my_specs, **kw)
    return gather_records(local, count)


# ------------------------------------------------------------------------------------------------------------------
# Real matrices (the f3 loaders: every OpenFOAM / StAn system comes from a file on rank 0): scatter of the CSR arrays
# and right-hand sides to their owners, optional gather of the solutions.
# ------------------------------------------------------------------------------------------------------------------
def _default_local_solver(items, *, precond: str = "jacobi", rtol_sq: float = 1e-8, atol_sq: float = 0.0, max_iter: int = 1024,
                          flags: int = 0, reorder: str | None = None):
    """Solve this rank's share one system after another on its GPU.  items: (rowptr, col, val, b) CUDA tensors.
    Returns (records (k, 4) = [iterations, status, final_res, seconds], [x tensors])."""
    from .operators import IC0, Jacobi
    recs = np.zeros((len(items), 4), dtype=np.float64)
    xs = []
    for i, (rp, ci, v, b) in enumerate(items):
        S = CsrSystem.from_any((rp, ci, v), reorder=reorder)
        S.set_preconditioner(Jacobi() if precond == "jacobi" else (IC0("solve") if precond == "ic0" else None))
        r = S.solve(b, rtol_sq=rtol_sq, atol_sq=atol_sq, max_iter=max_iter, flags=flags, want_history=False)
        recs[i] = (r.iterations, r.status, r.final_res, r.seconds)
        xs.append(r.x)
        S.close()
    return recs, xs


def scatter_systems(systems):
    """The scatter of REAL matrices: all ranks call it, rank 0 passes `systems` = [(rowptr int32, col
    int32, val fp64, b fp64), ...] (numpy arrays or torch tensors), the others None.  A size table (n, nnz per system) is
    broadcast; then, in rounds of one system per peer, rank 0 posts the four arrays of every peer's next system as ONE group
    of point-to-point sends (`batch_isend_irecv`: on the nccl backend a ncclGroupStart / ncclSend x 4 per peer /
    ncclGroupEnd, i.e. one xGMI link per peer, all peers at once) while each peer posts the matching receives into its own
    HBM.  Rank 0 keeps its own systems.  Returns (this rank's systems in `shard` order, their batch indices, (count, 2)
    size table)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _comm_device()

    def to_dev(a, dtype):
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        return t.to(device=dev, dtype=dtype).contiguous()

    meta = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        meta[0] = len(systems)
    dist.broadcast(meta, 0)
    count = int(meta.item())
    sizes = torch.zeros((count, 2), dtype=torch.int64, device=dev)
    if rank == 0 and count:
        sizes.copy_(torch.tensor([[len(s[0]) - 1, len(s[1])] for s in systems], dtype=torch.int64))
    dist.broadcast(sizes, 0)
    sizes_h = sizes.cpu().numpy()
    dtypes = (torch.int32, torch.int32, torch.float64, torch.float64)

    def shapes(s):
        n, nnz = int(sizes_h[s, 0]), int(sizes_h[s, 1])
        return (n + 1, nnz, nnz, n)

    mine: dict[int, tuple] = {}
    for base in range(0, count, world):                         # one round = one system per rank
        ops, staged = [], []
        for s in range(base, min(base + world, count)):
            owner = s % world
            if rank == 0:
                parts = tuple(to_dev(a, dt) for a, dt in zip(systems[s], dtypes))
                if owner == 0:
                    mine[s] = parts
                else:
                    staged.append(parts)                        # keep alive until the sends have completed
                    ops += [dist.P2POp(dist.isend, t, owner) for t in parts]
            elif owner == rank:
                parts = tuple(torch.empty(m, dtype=dt, device=dev) for m, dt in zip(shapes(s), dtypes))
                mine[s] = parts
                ops += [dist.P2POp(dist.irecv, t, 0) for t in parts]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        del staged
    order = shard(count, rank, world)
    return [mine[s] for s in order], order, sizes_h


def gather_solutions(xs, order, sizes_h):
    """Every owner sends its solutions back to rank 0 (grouped point-to-point, the scatter's form).  Returns the list of x in
    batch order on rank 0, None elsewhere."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _comm_device()
    count = len(sizes_h)
    out_x: list = [None] * count
    for base in range(0, count, world):
        ops, keep = [], []
        for s in range(base, min(base + world, count)):
            owner = s % world
            if owner == 0:
                if rank == 0:
                    out_x[s] = xs[order.index(s)]
            elif rank == owner:
                t = xs[order.index(s)].to(device=dev, dtype=torch.float64).contiguous()
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, 0))
            elif rank == 0:
                out_x[s] = torch.empty(int(sizes_h[s, 0]), dtype=torch.float64, device=dev)
                ops.append(dist.P2POp(dist.irecv, out_x[s], owner))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        del keep
    return out_x if rank == 0 else None


def solve_systems_distributed(systems, *, gather_x: bool = False, local_solver=None, **kw):
    """All ranks call this; rank 0 passes `systems` = [(rowptr int32, col int32, val fp64, b fp64), ...] as numpy
    arrays or torch tensors (what `io.load_case` / `coo_to_csr_device` return), the others None.

    scatter: `scatter_systems` (size table broadcast, then the arrays as grouped point-to-point sends, one xGMI link per
    peer).  solve: rank r owns systems r, r + world, ... and solves them with no communication.
    gather: fixed-size records are all-gathered (`gather_records`); with `gather_x` every owner sends its solutions back to
    rank 0 (`gather_solutions`).  Returns the (count, 4) record table on every rank -- and, on rank 0 with
    `gather_x`, `(table, [x_0, ..., x_{count-1}])` in batch order.
    `local_solver(items, **kw) -> (records, xs)` defaults to solving on this rank's GPU (tests inject a stand-in to
    run the exchange on CPU / gloo)."""
    import torch.distributed as dist
    solver = local_solver or _default_local_solver
    items, order, sizes_h = scatter_systems(systems)
    recs, xs = solver(items, **kw)
    del items
    table = gather_records(recs, len(sizes_h))
    if not gather_x:
        return table
    out_x = gather_solutions(xs, order, sizes_h)
    return (table, out_x) if dist.get_rank() == 0 else table
