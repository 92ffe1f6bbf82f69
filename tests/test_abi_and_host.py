"""CPU-only checks: the C-ABI library loads and exports every symbol include/dpcg.h declares, the
host-side input normalisation, and the multi-process batch sharding (gloo, world_size 2).
No compute call is made here: without a GPU the library must refuse, not fall back."""

import ctypes as C
import os
import pathlib
import re
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent

import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import _lib, batch, operators  # noqa: E402
from oracle import oracle as O  # noqa: E402  (test input generator)


def test_every_declared_symbol_is_exported_and_bound():
    header = (ROOT / "include" / "dpcg.h").read_text()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(dpcg_[a-z0-9_]+)\(", header, flags=re.M))
    assert len(declared) >= 24
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in dpcg.h but not exported by libdpcg.so"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dpcg_version() >= 100
    assert lib.dpcg_status_string(0) == b"ok"


def test_header_cites_the_reference_for_every_entry_point_group():
    header = (ROOT / "include" / "dpcg.h").read_text()
    for cite in ("cg.py:50-90", "cg.py:20-47", "cg.py:61,81", "test.py:70-72", "test.py:74-79", "test.py:83",
                 "utils.py:15-43", "utils.py:66-72"):
        assert cite in header, cite


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    lib = _lib.lib()
    h = C.c_void_p()
    rp = np.array([0, 1], dtype=np.int32)
    ci = np.array([0], dtype=np.int32)
    v = np.array([2.0])
    st = lib.dpcg_create(C.byref(h), 1, 1, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                         v.ctypes.data_as(C.c_void_p), _lib.F64, _lib.HOST, 1, None)
    assert st == _lib.ERR_HIP
    with pytest.raises(_lib.DpcgError):
        _lib.check(st)
    from deeppreconditioning_amd.cg import preconditioned_conjugate_gradient
    with pytest.raises(Exception):  # no device: the drop-in raises instead of computing on the CPU
        preconditioned_conjugate_gradient(sp.eye(4, format="csr"), torch.ones(4, dtype=torch.float64), None)


def test_product_never_imports_the_oracle():
    """The checker stays a checker: nothing under the package imports, links or calls oracle/."""
    needles = ("import oracle", "from oracle", "liboracle", "pcg_oracle", "orc_", "c_oracle")
    for path in (ROOT / "deeppreconditioning_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h") or path.name == "Makefile":
            text = path.read_text()
            for needle in needles:
                assert needle not in text, f"{path} references {needle}"


def test_poisson_sizes_and_argument_checks():
    lib = _lib.lib()
    rows, nnz = C.c_int64(), C.c_int64()
    assert lib.dpcg_poisson_sizes(2, 1024, C.byref(rows), C.byref(nnz)) == 0
    assert (rows.value, nnz.value) == (1048576, 5238784)
    assert lib.dpcg_poisson_sizes(3, 100, C.byref(rows), C.byref(nnz)) == 0
    assert (rows.value, nnz.value) == (1000000, 6940000)
    assert lib.dpcg_poisson_sizes(3, 256, C.byref(rows), C.byref(nnz)) == 0
    assert (rows.value, nnz.value) == (16777216, 117047296)
    assert lib.dpcg_poisson_sizes(4, 10, C.byref(rows), C.byref(nnz)) == _lib.ERR_INVALID
    assert lib.dpcg_poisson_sizes(3, 1000, C.byref(rows), C.byref(nnz)) == _lib.ERR_INVALID  # int32 CSR limit
    assert lib.dpcg_spmv(None, None, None, None) == _lib.ERR_INVALID
    assert b"NULL" in lib.dpcg_last_error()


def test_csr_arrays_normalisation():
    A = sp.random(30, 30, density=0.2, random_state=0, format="coo")
    A = (A + A.T + 30 * sp.eye(30)).tocoo()
    ref = A.tocsr()
    ref.sort_indices()
    for obj in (A, A.tocsr(), A.toarray(), torch.from_numpy(A.toarray()),
                torch.from_numpy(A.toarray()).to_sparse_csr(), torch.from_numpy(A.toarray()).to_sparse()):
        space, rp, ci, v, n = operators.csr_arrays(obj)
        assert space == "host" and n == 30
        assert rp.dtype == np.int32 and ci.dtype == np.int32 and v.dtype == np.float64
        assert np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices)
        np.testing.assert_allclose(v, ref.data)
    dup = sp.coo_matrix((np.ones(3), ([0, 0, 1], [0, 0, 1])), shape=(2, 2))  # duplicates are summed
    _, rp, ci, v, _ = operators.csr_arrays(dup)
    assert list(rp) == [0, 1, 2] and list(v) == [2.0, 1.0]
    with pytest.raises(ValueError):
        operators.csr_arrays(np.ones((3, 4)))
    with pytest.raises(TypeError):
        operators.csr_arrays("not a matrix")


def test_preconditioner_mapping():
    n = 12
    assert isinstance(operators.as_preconditioner(None, n), D.Identity)
    d = np.linspace(1, 2, n)
    diag_csr = torch.sparse_coo_tensor(torch.vstack((torch.arange(n), torch.arange(n))), torch.from_numpy(d),
                                       size=(n, n)).to_sparse_csr()        # what test.py:74-79 builds
    pc = operators.as_preconditioner(diag_csr, n)
    assert isinstance(pc, D.Jacobi) and np.array_equal(np.asarray(pc.dinv), d)
    assert isinstance(operators.as_preconditioner(torch.from_numpy(np.diag(d)), n), D.Jacobi)  # dense, train.py:100
    M = sp.random(n, n, density=0.3, random_state=1, format="csr") + sp.eye(n)
    assert isinstance(operators.as_preconditioner(M, n), D.CsrPreconditioner)
    spec = D.LLtSolve(sp.tril(M, format="csr"))
    assert operators.as_preconditioner(spec, n) is spec
    with pytest.raises(TypeError):
        operators.as_preconditioner(object(), n)

    class Duck:                                    # all cg.py:61,81 ask of M: `M @ rk`
        def __matmul__(self, r):
            return r

    assert isinstance(operators.as_preconditioner(Duck(), n), D.OperatorPreconditioner)
    with pytest.raises(ValueError):
        operators.as_preconditioner(sp.eye(n + 1, format="csr"), n)
    with pytest.raises(ValueError):
        D.IC0("bogus")


def test_shard_partition():
    for count in (0, 1, 7, 64):
        for world in (1, 2, 8):
            parts = [batch.shard(count, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(count))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert batch.shard(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]  # BASELINE config 4: 8 systems per GPU


def test_bench_algorithmic_bytes():
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.spmv_bytes(1048576, 5238784) == 83836932
    assert bench.spmv_bytes(1000000, 6940000) == 103280004
    assert bench.spmv_bytes(16777216, 117047296) == 1740111876


# ---- world_size-2 gloo run of the scatter / shard / gather path (no GPU: the local solver is a stand-in) ----
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_local_solver(specs, **kw):
    # record = [iterations, status, final_res, seconds] derived from the spec so the gather order is checkable
    return np.array([[sp_.n * 10 + sp_.dim, 0, 1.0 / (1 + sp_.seed), 0.5] for sp_ in specs], dtype=np.float64).reshape(-1, 4)


def _worker(rank, world, port, count, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [batch.SystemSpec(2 + (i % 2), 10 + i, i) for i in range(count)] if rank == 0 else None
    out = batch.solve_specs_distributed(specs, local_solver=_fake_local_solver)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("count", [5, 8])
def test_distributed_scatter_shard_gather_gloo_world2(count):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, count, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _fake_local_solver([batch.SystemSpec(2 + (i % 2), 10 + i, i) for i in range(count)])
    for rank in (0, 1):  # every rank holds the full table in batch order
        np.testing.assert_array_equal(results[rank], expect)


def _worker_config4(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []

    def local_solver(specs, **kw):
        seen.extend((sp_.dim, sp_.n, sp_.seed) for sp_ in specs)
        return _fake_local_solver(specs)

    specs = [batch.SystemSpec(3, 256, s) for s in range(64)] if rank == 0 else None
    out = batch.solve_specs_distributed(specs, local_solver=local_solver)
    q.put((rank, out, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_config4_shape_gloo_world8():
    """BASELINE config 4's exact shape -- 64 systems poisson3d(256), 8 ranks, 8 per rank -- through scatter / shard / gather over gloo
    with a stand-in local solver (no GPU here): system s lands on rank s mod 8 with its own seed, every rank ends with the full
    64-row table in batch order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_config4, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = {r: (t, seen) for r, t, seen in (q.get(timeout=300) for _ in range(8))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = _fake_local_solver([batch.SystemSpec(3, 256, s) for s in range(64)])
    for rank in range(8):
        table, seen = results[rank]
        np.testing.assert_array_equal(table, expect)
        assert seen == [(3, 256, s) for s in range(rank, 64, 8)]              # 8 per rank, system s on rank s mod 8


# ---- the same for REAL matrices: the CSR arrays and right-hand sides travel, the solutions come back -------------
def _fake_matrix_solver(items, **kw):
    """Stand-in for the GPU solve: x = A b computed from the arrays AS RECEIVED, so a corrupted or misrouted matrix shows
    up in the gathered solutions; record = [n, nnz, sum(val), sum(b)]."""
    recs, xs = [], []
    for rp, ci, v, b in items:
        A = sp.csr_matrix((v.numpy(), ci.numpy(), rp.numpy()), shape=(len(rp) - 1,) * 2)
        xs.append(torch.from_numpy(A @ b.numpy()))
        recs.append([A.shape[0], A.nnz, float(v.numpy().sum()), float(b.numpy().sum())])
    return np.array(recs, dtype=np.float64).reshape(-1, 4), xs


def _real_systems(count):
    out = []
    for i in range(count):
        A = O.unstructured_like(O.poisson2d(5 + i), seed=i) if i % 2 else O.poisson2d(5 + i)
        out.append((A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), O.rhs(A.shape[0], i)))
    return out


def _worker_real(rank, world, port, count, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    systems = _real_systems(count) if rank == 0 else None
    out = batch.solve_systems_distributed(systems, gather_x=True, local_solver=_fake_matrix_solver)
    if rank == 0:
        table, xs = out
        q.put((rank, table, [x.numpy() for x in xs]))
    else:
        q.put((rank, out, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("count", [1, 5])
def test_distributed_real_matrix_scatter_and_x_gather_gloo_world2(count):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, count, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {r: (t, x) for r, t, x in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    systems = _real_systems(count)
    expect = np.array([[len(rp) - 1, len(ci), v.sum(), b.sum()] for rp, ci, v, b in systems])
    np.testing.assert_array_equal(results[0][0], expect)            # records in batch order on every rank
    np.testing.assert_array_equal(results[1][0], expect)
    for (rp, ci, v, b), x in zip(systems, results[0][1]):           # x of every system is back on rank 0, bit for bit
        A = sp.csr_matrix((v, ci, rp), shape=(len(rp) - 1,) * 2)
        np.testing.assert_array_equal(x, A @ b)
