"""Host-side label tensors (the step just before the hot path).

  encode_boxes      utils/tools.py:179-209   pixel boxes -> finest-grid label (last writer wins)
  down2xlabel       utils/tools.py:342-367   label pyramid for the coarser FPN levels
  get_class_weight  utils/tools.py:592-627   e.g. the `binary` positive/negative ratio
  label_pyramid     yolov3/__init__.py:41-53 (_Yolov3DataSequence): [coarse, ..., fine]

Plain NumPy, vectorised where the reference loops in Python; results are identical to the
reference's (tests/test_labels_cpu.py checks them against the golden vectors).
"""
import numpy as np


def encode_boxes(boxes, labels, img_hw, grid_shape, class_num):
    """boxes: iterable of (x1, y1, x2, y2) in pixels; labels: class ids. Returns (gh, gw, 5+C) float64."""
    label = np.zeros((grid_shape[0], grid_shape[1], 5 + class_num))
    img_h, img_w = img_hw
    cell_h, cell_w = img_h / grid_shape[0], img_w / grid_shape[1]
    for (x1, y1, x2, y2), lab in zip(boxes, labels):
        bx, by, bw, bh = x1 + (x2 - x1) / 2, y1 + (y2 - y1) / 2, x2 - x1, y2 - y1
        x_i, y_i = int(bx // cell_w), int(by // cell_h)
        if x_i < grid_shape[1] and y_i < grid_shape[0]:
            label[y_i, x_i, 0] = bx % cell_w / cell_w
            label[y_i, x_i, 1] = by % cell_h / cell_h
            label[y_i, x_i, 2] = bw / img_w
            label[y_i, x_i, 3] = bh / img_h
            label[y_i, x_i, 4] = 1
            label[y_i, x_i, 5 + lab] = 1
    return label


def down2xlabel(label_data):
    """Downsample label by 2x: per 2x2 block that holds an object keep the box with the largest w*h,
    re-express its centre offset in the coarser cell."""
    label_data = np.asarray(label_data)
    b, gh, gw, ch = label_data.shape
    h2, w2 = gh // 2, gw // 2
    blocks = label_data[:, :h2 * 2, :w2 * 2].reshape(b, h2, 2, w2, 2, ch).transpose(0, 1, 3, 2, 4, 5)
    blocks = blocks.reshape(b, h2, w2, 4, ch)                      # index = dy*2 + dx, as crop[...].argmax()
    has = blocks[..., 4].max(axis=-1) == 1
    max_id = (blocks[..., 2] * blocks[..., 3]).argmax(axis=-1)      # first maximum, like ndarray.argmax
    pick = np.take_along_axis(blocks, max_id[..., None, None], axis=3)[..., 0, :]
    off = np.stack([max_id % 2, max_id // 2], axis=-1)
    new_label = np.zeros((b, h2, w2, ch))
    new_xy = (pick[..., :2] + off) / 2
    new_label[..., :2] = np.where(has[..., None], new_xy, 0)
    new_label[..., 2:] = np.where(has[..., None], pick[..., 2:], 0)
    return new_label


def label_pyramid(label_data, levels):
    """[coarsest, ..., finest] as _Yolov3DataSequence builds it (yolov3/__init__.py:47-53)."""
    out = [label_data]
    for _ in range(levels - 1):
        label_data = down2xlabel(label_data)
        out.insert(0, label_data)
    return out


def label_pyramid_device(boxes_per_image, classes_per_image, img_hw, finest_grid, class_num, levels):
    """The label tensors of one batch built ON THE DEVICE (csrc/labels.hip): boxes_per_image = list (one entry per
    image) of [k,4] arrays (x1, y1, x2, y2 in pixels), classes_per_image = list of [k] class ids. Returns the float32
    CUDA tensors [coarsest, ..., finest] the losses consume -- bit-identical to label_pyramid(encode_boxes(...)) cast
    to float32 (tests/test_gpu_labels.py). Only the few KB of box coordinates cross PCIe, not the 19 MB of labels a
    416x416 C=80 batch of 32 needs per step and rank."""
    import torch
    from . import ops
    n = len(boxes_per_image)
    first = np.zeros(n + 1, dtype=np.int32)
    first[1:] = np.cumsum([len(b) for b in boxes_per_image])
    boxes = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 4) for b in boxes_per_image] or [np.zeros((0, 4))])
    cls = np.concatenate([np.asarray(c, dtype=np.int32).reshape(-1) for c in classes_per_image] or [np.zeros(0, np.int32)])
    if len(boxes) == 0:
        boxes, cls = np.zeros((1, 4)), np.zeros(1, np.int32)     # (never read: every range is empty)
    l64, l32 = ops.encode_labels(torch.from_numpy(boxes).cuda(), torch.from_numpy(cls.astype(np.int32)).cuda(),
                                 torch.from_numpy(first).cuda(), n, img_hw, finest_grid, class_num)
    out = [l32]
    for _ in range(levels - 1):
        l64, l32 = ops.down2xlabel(l64)
        out.insert(0, l32)
    return out


def get_class_weight(label_data, method="alpha"):
    label_data = np.asarray(label_data)
    total = int(np.prod(label_data.shape[:-1]))
    samples = np.array([label_data[..., i].sum() for i in range(label_data.shape[-1])])
    if method == "effective":
        beta = (total - 1) / total
        w = (1 - beta) / (1 - np.power(beta, samples))
    elif method == "binary":
        return samples / (total - samples)
    else:
        w = 1 / samples
    if method == "log":
        w = np.log(total * w)
    return w / np.sum(w) * len(w)


def synthetic_batch(rng, N, input_hw, class_num, levels=3, finest_stride=8, max_boxes=8):
    """SURVEY.md section 8d synthetic data: images U[0,1); per image k~U{1..8} boxes, centre ~U(0,1)^2,
    w,h ~U(.05,.6), class ~U{0..C-1}; encoded on the finest grid, then the label pyramid."""
    H, W = input_hw
    x = rng.random((N, H, W, 3), dtype=np.float32)
    gh, gw = H // finest_stride, W // finest_stride
    fine = np.zeros((N, gh, gw, 5 + class_num))
    for n in range(N):
        k = int(rng.integers(1, max_boxes + 1))
        cx, cy = rng.random(k) * W, rng.random(k) * H
        bw, bh = rng.uniform(.05, .6, k) * W, rng.uniform(.05, .6, k) * H
        boxes = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], axis=1)
        fine[n] = encode_boxes(boxes, rng.integers(0, class_num, k), (H, W), (gh, gw), class_num)
    return x, [a.astype(np.float32) for a in label_pyramid(fine, levels)]


def synthetic_keras_weights(builder, seed, residual_gamma=1.0):
    """SURVEY.md section 8d synthetic weights, keyed like Model.get_layer(name).get_weights(): '<layer>/<i>' -> float32 array in
    KERAS layout (conv kernels HWIO). He-normal N(0, 2 / fan_in) kernels, zero biases, BatchNormalization gamma 1 / beta 0 /
    moving mean 0 / moving variance 1. Drawn from ONE numpy generator in graph order, so that the GPU model (bench.py's C5
    block loads them with set_weights) and the CPU side (tests/golden/make_timing.py: the oracle's forward feeding the
    reference's decode / NMS; bench.py's cpu_baseline) hold the same network without exchanging a file."""
    rng = np.random.default_rng(seed)
    w = {}
    for u in builder.units:
        if u.kind == "conv":
            cin = u.src.c
            w[f"{u.name}_conv/0"] = (rng.standard_normal((u.k, u.k, cin, u.cout)) * (2.0 / (u.k * u.k * cin)) ** 0.5).astype(np.float32)
            if u.bias:
                w[f"{u.name}_conv/1"] = np.zeros(u.cout, np.float32)
            if u.bn:
                gam = residual_gamma if u.residual is not None else 1.0
                w[f"{u.name}_bn/0"], w[f"{u.name}_bn/1"] = np.full(u.cout, gam, np.float32), np.zeros(u.cout, np.float32)
                w[f"{u.name}_bn/2"], w[f"{u.name}_bn/3"] = np.zeros(u.cout, np.float32), np.ones(u.cout, np.float32)
        elif u.kind == "head":
            cin = u.src.c
            for j in range(u.A):
                for part, c in (("xy", 2), ("wh", 2), ("conf", 1), ("prob", u.C)):
                    w[f"out{u.level + 1}_box{j + 1}_{part}_conv/0"] = (rng.standard_normal((1, 1, cin, c)) * (2.0 / cin) ** 0.5).astype(np.float32)
                    w[f"out{u.level + 1}_box{j + 1}_{part}_conv/1"] = np.zeros(c, np.float32)
    return w
