"""Child process of test_nccl_world1_runs_the_scatter_and_gather_on_rccl: one rank, backend "nccl" (= RCCL on ROCm), so
that broadcast / grouped send-recv / all_gather of deeppreconditioning_amd.batch execute on RCCL at least once.  Started
before anything touches the GPU in this process; prints one JSON line."""
import json
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                            device_id=torch.device("cuda", 0))
    import deeppreconditioning_amd as D
    from deeppreconditioning_amd import batch
    from oracle import oracle as O
    specs = [batch.SystemSpec(2, 64, 0), batch.SystemSpec(3, 16, 1), batch.SystemSpec(2, 256, 0)]
    table = batch.solve_specs_distributed(specs)
    mats = [O.poisson2d(40), O.unstructured_like(O.poisson3d(12), seed=1), O.poisson2d(90)]
    systems = [(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), O.rhs(A.shape[0], i))
               for i, A in enumerate(mats)]
    table2, xs = batch.solve_systems_distributed(systems, gather_x=True)
    same = []
    for (rp, ci, v, b), x in zip(systems, xs):
        S = D.CsrSystem.from_any((rp, ci, v))
        S.set_preconditioner(D.Jacobi())
        same.append(bool(torch.equal(S.solve(torch.from_numpy(b).cuda()).x, x)))
        S.close()
    print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "spec_table": table.tolist(),
                      "real_table": table2.tolist(), "x_equal_to_direct_solve": same}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
