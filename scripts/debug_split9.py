import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, K, C = 1, 8, 64, 32
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
sm = lambda s: torch.randint(-3, 4, s).float()
def chk(name, x, w):
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double().reshape(-1, K)
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double()).reshape(-1, K)
    diff = y - ref
    print(f"{name:34s} wrong {(diff != 0).float().mean().item():.3f} sample {diff[diff != 0][:6].tolist()}")
shape = (N, H, H, C)
chk("x even 128..254, w small", torch.randint(64, 128, shape).float() * 2, sm((K, C)))
chk("x odd 129..255, w small", torch.randint(64, 128, shape).float() * 2 + 1, sm((K, C)))
chk("x odd 129..255 positive, w +-1", torch.randint(64, 128, shape).float() * 2 + 1, torch.randint(0, 2, (K, C)).float() * 2 - 1)
chk("x = 129 const, w small", torch.full(shape, 129.0), sm((K, C)))
chk("x = 255 const, w small", torch.full(shape, 255.0), sm((K, C)))
chk("x = 255 const, w in {0,1}", torch.full(shape, 255.0), torch.randint(0, 2, (K, C)).float())
chk("x = 1 const, w small", torch.full(shape, 1.0), sm((K, C)))
chk("x = 127 const, w small", torch.full(shape, 127.0), sm((K, C)))
w1 = torch.zeros(K, C); w1[:, 0] = 1; w1[:, 17] = 1
chk("x odd, w picks ch 0 and 17", torch.randint(64, 128, shape).float() * 2 + 1, w1)
w2 = torch.zeros(K, C); w2[:, 0] = 1; w2[:, 17] = -1
chk("x odd, w = +ch0 -ch17", torch.randint(64, 128, shape).float() * 2 + 1, w2)
w3 = torch.zeros(K, C); w3[:, 3] = 1; w3[:, 5] = -1
chk("x odd, w = +ch3 -ch5 (same stage)", torch.randint(64, 128, shape).float() * 2 + 1, w3)
print("---- negatives")
odd = lambda: torch.randint(64, 128, shape).float() * 2 + 1
sgn = torch.randint(0, 2, shape).float() * 2 - 1
chk("x odd negative, w = 1", -odd(), torch.ones(K, C))
chk("x odd negative, w small", -odd(), sm((K, C)))
chk("x odd mixed sign, w = 1", odd() * sgn, torch.ones(K, C))
chk("x odd mixed sign, w small", odd() * sgn, sm((K, C)))
chk("x even mixed sign, w small", (odd() - 1) * sgn, sm((K, C)))
chk("x odd mixed sign <=127, w small", (torch.randint(0, 64, shape).float() * 2 + 1) * sgn, sm((K, C)))
chk("x mixed 255/-255, w small", 255.0 * sgn, sm((K, C)))
chk("x mixed 129/-129, w = 1", 129.0 * sgn, torch.ones(K, C))
