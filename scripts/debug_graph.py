import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, yolov3
A9 = [[0.05,0.06],[0.08,0.1],[0.12,0.2],[0.2,0.15],[0.25,0.3],[0.3,0.45],[0.5,0.4],[0.6,0.7],[0.9,0.85]]
y = yolov3.Yolo((64, 64, 3), ["a", "b"]); y.create_model(anchors=A9, pretrained_body=None)
net = y.model.net
rng = np.random.default_rng(0)
x1 = torch.from_numpy(rng.random((2, 64, 64, 3), dtype=np.float32)).cuda()
g = [o.clone() for o in net.infer(x1)]
e = [o.clone() for o in net.forward(x1, training=False)]
print([float((a-b).abs().max()) for a,b in zip(g,e)])
p = y.model.predict(x1.cpu().numpy())
print([float(np.abs(a-b.cpu().numpy()).max()) for a,b in zip(p,e)], [a.shape for a in p])
p2 = y.model.predict(x1.cpu().numpy())
print([float(np.abs(a-b).max()) for a,b in zip(p,p2)])
