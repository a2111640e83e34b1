import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
N, H, C, K = 1, 8, 32, 64
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
def two(c1, a1, b1, c2, a2, b2):
    x = torch.zeros(N, H, H, C); w = torch.zeros(K, C)
    x.reshape(-1, C)[3, c1] = a1; w[7, c1] = b1
    x.reshape(-1, C)[3, c2] = a2; w[7, c2] = b2
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().reshape(-1, K)
    return y[3, 7].item(), a1 * b1 + a2 * b2
for (c1, c2) in ((0, 1), (0, 8), (0, 16), (3, 20), (7, 9)):
    for (a1, b1, a2, b2) in ((129, 129, 1, 1), (129, 129, 129, 129), (255, 255, 255, 255), (255, 3, 3, 255), (129, 1, 1, 129), (255, 255, -255, 255)):
        got, ref = two(c1, float(a1), float(b1), c2, float(a2), float(b2))
        flag = "" if got == ref else "   <-- WRONG"
        print(f"c=({c1},{c2}) {a1}*{b1} + {a2}*{b2}: got {got} ref {ref}{flag}")
