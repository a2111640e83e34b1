"""Numerical side of VERDICT r04 next #5 (scripts/hip_probe/wino_probe.cpp is the rate side): Winograd F(2x2, 3x3) on the
fp16 x 2-plane operand format against float64 -- NumPy emulation of the arithmetic a kernel would do:
  input  d  -> planes (h, l) of s_d * d (s_d = power of two from max|d|), rebuilt as float32 h + l
  V = B^T d B in float32, x 1/4, re-split into planes (h', l')
  U = G g G^T in float64 (done once per set of weights), scaled, split into planes
  M = three fp16 x fp16 products per pair (l*h + h*l + h*h), accumulated in float32 over the input channels
  Y = A^T M A in float32, unscaled
beside the DIRECT 3x3 convolution on the same planes (what conv_win_kernel computes). Relative error = max |Y - Y64| / max |Y64|."""
import numpy as np
rng = np.random.default_rng(3)
H = W = 16; CIN = 128; COUT = 64


def split(x, bound):
    s = 2.0 ** np.floor(np.log2(2.0 ** 15 / bound))
    h = (x * s).astype(np.float16)
    l = (x * s - h.astype(np.float64)).astype(np.float16)
    return h, l, s


def mac3(ah, al, bh, bl):
    """sum_k of the three plane products, float32 accumulation (one float32 matmul per pass, passes added smallest first)"""
    f = lambda a, b: (a.astype(np.float32) @ b.astype(np.float32))
    return f(al, bh) + f(ah, bl) + f(ah, bh)


x = rng.standard_normal((H + 2, W + 2, CIN)); x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0     # zero border = 'same' padding
w = rng.standard_normal((3, 3, CIN, COUT)) / np.sqrt(9 * CIN)
# float64 reference
ref = np.zeros((H, W, COUT))
for r in range(3):
    for s_ in range(3):
        ref += x[r:r + H, s_:s_ + W] @ w[r, s_]
scale = np.abs(ref).max()
xh, xl, sx = split(x, np.abs(x).max())
# ---- direct on planes ----
wh, wl, sw = split(w, np.abs(w).max())
yd = np.zeros((H, W, COUT), dtype=np.float32)
for r in range(3):
    for s_ in range(3):
        a_h, a_l = xh[r:r + H, s_:s_ + W].reshape(-1, CIN), xl[r:r + H, s_:s_ + W].reshape(-1, CIN)
        yd += mac3(a_h, a_l, wh[r, s_], wl[r, s_]).reshape(H, W, COUT)
yd = yd.astype(np.float64) / (sx * sw)
# ---- Winograd F(2x2, 3x3) on planes ----
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float32)
U = np.einsum("ir,rsco,js->ijco", G, w, G)                       # [4][4][CIN][COUT], float64 (per weight update)
uh, ul, su = split(U, np.abs(U).max())
xs = xh.astype(np.float32) + xl.astype(np.float32)                # rebuilt scaled input (exact)
yw = np.zeros((H, W, COUT), dtype=np.float32)
tiles = [(ty, tx) for ty in range(0, H, 2) for tx in range(0, W, 2)]
V = np.stack([np.einsum("ia,abc,jb->ijc", BT, xs[ty:ty + 4, tx:tx + 4], BT) for ty, tx in tiles])   # [T][4][4][CIN] float32
V = (V * np.float32(0.25)).astype(np.float32)
vh = V.astype(np.float16)
vl = (V - vh.astype(np.float32)).astype(np.float16)
M = np.zeros((len(tiles), 4, 4, COUT), dtype=np.float32)
for i in range(4):
    for j in range(4):
        M[:, i, j] = mac3(vh[:, i, j], vl[:, i, j], uh[i, j], ul[i, j])
Y = np.einsum("ia,tabo,jb->tijo", AT, M, AT)                      # float32
for t, (ty, tx) in enumerate(tiles):
    yw[ty:ty + 2, tx:tx + 2] = Y[t]
yw = yw.astype(np.float64) * 4.0 / (sx * su)
print(f"3x3 'same' conv {H}x{W}, {CIN} -> {COUT} channels, random normal data; error relative to max|y| = {scale:.3f}")
print(f"  direct on planes (3 passes):            {np.abs(yd - ref).max() / scale:.3e}")
print(f"  Winograd F(2x2,3x3) on planes (3 passes): {np.abs(yw - ref).max() / scale:.3e}")
