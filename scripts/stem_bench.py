import os, sys, torch
sys.path.insert(0, '/root/repo')
from tf2_yolo_amd import ops
d = ops.conv_desc((32, 416, 416, 3), 32, 3, 3, 1, "same")
x = torch.rand(32, 416, 416, 3, device="cuda"); w = torch.randn(32, 27, device="cuda") * 0.2
y = torch.empty(32, 416, 416, 32, device="cuda")
st = torch.zeros(ops.BN_STAT_SLOTS * 64, device="cuda", dtype=torch.float64); am = torch.zeros(32, device="cuda", dtype=torch.int32)
def run(): ops.conv2d_fwd(d, x, w, None, out=y, stats=st, absmax=am)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("YOLO_STEM_SREG", "default"), "stem fwd us", e0.elapsed_time(e1) / 20 * 1e3, "checksum", float(y.double().sum()))
