"""PreconditionerNet forward at BASELINE config 2's size (256^2 5-point system, channels 1-16-32-64-32-16-1, seeded random weights), plan
cached, 20 times -- the workload of `rocprofv3 --kernel-trace --stats` for profiles/r04_cnn_kernel_stats.csv.  Prints the flop / byte
model of every layer (deeppreconditioning_amd.model.forward_cost) beside the event-timed forward."""
import json
import sys

import numpy as np
import scipy.sparse as sp
import torch

from deeppreconditioning_amd import model as mdl

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(69)
net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
idx = np.arange(n2 * n2)
A2 = sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)],
              [0, -1, -n2], format="csr")
inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")
with torch.no_grad():
    for _ in range(3):
        net(inp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net(inp)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
cost = mdl.forward_cost(net, inp)
print(json.dumps({"forward_ms": round(ms, 4), "tflops": round(cost["flops"] / ms / 1e9, 2), "cost": cost}, indent=1))
