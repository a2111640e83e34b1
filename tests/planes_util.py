"""Test helper: decode a planes buffer (tf2_yolo_amd/csrc/planes.hpp) back to a dense float64 matrix."""
import torch


def planes_to_dense(pl, rows, c):
    """pl: uint8 tensor (CPU). Returns (x [rows, c] float64 = (h + l) / s, bound, scale, tail_is_zero)."""
    nblk = (rows + 15) // 16
    body = (nblk + 1) * (c // 16) * 1024
    hdr = pl[body:body + 12].clone().view(torch.float32)
    u = pl[:body].clone().view(torch.float16).reshape(nblk + 1, c // 16, 2, 2, 16, 8)   # [blk][kb][plane][half][row][8]
    f = u.permute(2, 0, 4, 1, 3, 5).reshape(2, (nblk + 1) * 16, c).double()            # [plane][row][channel]
    dense = (f[0] + f[1]) / float(hdr[1])
    return dense[:rows], float(hdr[0]), float(hdr[1]), bool((f[:, rows:] == 0).all())
