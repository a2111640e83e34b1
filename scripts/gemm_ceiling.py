"""Library fp16 GEMM rates on this box (random and all-zero operands) and the bare-MFMA probe: the practical MFMA ceiling
that the planes conv kernels' raw MFMA rate (3 passes per product) is compared with in DESIGN.md section 3.1.
usage: python scripts/gemm_ceiling.py [out.json]"""
import json
import os
import socket
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda:0"


def rate(M, N, K, zeros, dtype=torch.float16, iters=30):
    a = (torch.zeros if zeros else torch.randn)(M, K, device=dev, dtype=dtype)
    b = (torch.zeros if zeros else torch.randn)(K, N, device=dev, dtype=dtype)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms


rows = []
for (M, N, K) in [(8192, 8192, 8192), (86528, 256, 1152), (86528, 128, 2304), (21632, 512, 2304), (5408, 1024, 4608),
                  (86528, 256, 128)]:
    for z in (0, 1):
        tf, ms = rate(M, N, K, z)
        rows.append({"M": M, "N": N, "K": K, "operands": "zeros" if z else "random normal", "tflops": round(tf, 1),
                     "us": round(ms * 1e3, 1)})
        print(rows[-1], flush=True)
from tf2_yolo_amd import ops
probe = ops.mfma_ceiling(0.2)
print("bare MFMA probe:", probe)
try:
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
    smi = json.loads(smi) if smi.strip().startswith("{") else smi[:2000]
except Exception as e:
    smi = repr(e)
out = {"what": "torch.matmul (hipBLASLt) fp16 GEMM rates and the bare v_mfma_f32_32x32x16_f16 loop of csrc/probe.hip on one box",
       "device": torch.cuda.get_device_name(0), "host": socket.gethostname(), "torch": torch.__version__,
       "gemm": rows, "bare_mfma_probe_random_fp16": probe, "rocm_smi_after": smi}
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
