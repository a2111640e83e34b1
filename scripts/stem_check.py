import os, sys, torch
sys.path.insert(0, '/root/repo')
from tf2_yolo_amd import ops
torch.manual_seed(0)
N=8
d = ops.conv_desc((N, 416, 416, 3), 32, 3, 3, 1, "same")
x = torch.rand(N, 416, 416, 3, device="cuda"); w = torch.randn(32, 27, device="cuda") * 0.2
b = torch.randn(32, device="cuda")
y = torch.empty(N, 416, 416, 32, device="cuda")
st = torch.zeros(ops.BN_STAT_SLOTS * 64, device="cuda", dtype=torch.float64); am = torch.zeros(32, device="cuda", dtype=torch.int32)
ops.conv2d_fwd(d, x, w, b, out=y, stats=st, absmax=am)
ref = torch.nn.functional.conv2d(x.permute(0,3,1,2).double(), w.reshape(32,3,3,3).permute(0,3,1,2).double(), b.double(), padding=1).permute(0,2,3,1)
err = (y.double()-ref).abs().max().item()/ref.abs().max().item()
print(os.environ.get("YOLO_STEM_SREG","default"), "relerr", err, "stats", float(st.reshape(64,2,32).sum(0)[0].sum()), float(ref.sum()))
