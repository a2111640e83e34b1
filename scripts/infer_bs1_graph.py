"""YOLOv3-416 bs-1 inference through the captured hipGraph (Model.predict's path): ms per forward.
usage: infer_bs1_graph.py [option_key=value ...]   (yolo_set_option overrides, e.g. 2=0 switches split-K off)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import yolov3
from tf2_yolo_amd import ops


def run(tag):
    y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
    y.create_model(pretrained_body=None)
    x = torch.from_numpy(np.random.default_rng(1234).random((1, 416, 416, 3), dtype=np.float32)).cuda()
    net = y.model.net
    for _ in range(3):
        net.infer(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            net.infer(x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    print(tag, "graph replay ms", round(best, 3), flush=True)


if __name__ == "__main__":
    sets = [a for a in sys.argv[1:]] or [""]
    for spec in sets:
        ops.reset_options()
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            ops.set_option(int(k), int(v))
        run(spec or "defaults")
