"""rocprofv3 --pmc FETCH_SIZE target: three weight-gradient launches of cad_gemm_stream at the configs[2] shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
a = torch.empty(1024, 262144, dtype=torch.bfloat16, device=dev).normal_()
b = torch.empty(262144, 256, dtype=torch.bfloat16, device=dev).normal_()
for _ in range(3):
    ops.wgrad_cm_tm(a, b)
torch.cuda.synchronize()
