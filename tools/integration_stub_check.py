import ctypes as C, torch, sys
sys.path.insert(0, "/root/repo")
lib = C.CDLL("/root/repo/deeppreconditioning_amd/csrc/libdpcg.so")
lib.dpcg_create.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.dpcg_set_precond_jacobi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
lib.dpcg_destroy.argtypes = [C.c_void_p]
lib.dpcg_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                           C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.c_void_p, C.c_void_p, C.c_void_p]

def pcg(rowptr, col, val, b, rtol=1e-8, max_iter=1024):          # int32, int32, float64 CUDA tensors
    h, iters, res, sec = C.c_void_p(), C.c_int(), C.c_double(), C.c_double()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, nnz = b.numel(), col.numel()
    assert lib.dpcg_create(C.byref(h), n, nnz, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), 0, 0, 0, stream) == 0
    assert lib.dpcg_set_precond_jacobi(h, None, 0, stream) == 0   # M = diag(1/a_ii), test.py:74-79
    x = torch.empty_like(b)
    status = lib.dpcg_solve(h, b.data_ptr(), None, x.data_ptr(), rtol, 0.0, max_iter, 0, stream,
                            C.byref(iters), C.byref(res), C.byref(sec), None, None, None)
    lib.dpcg_destroy(h)
    return sec.value, iters.value, status

from deeppreconditioning_amd import poisson
rp, ci, v = poisson.poisson_csr(2, 256)
print(pcg(rp, ci, v, poisson.rhs(256 * 256, 0)))
