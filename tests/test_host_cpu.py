"""CPU-only checks of the host logic: C-ABI library exports, graph builders (architecture
cross-checks against the published Darknet / keras-yolo3 figures), facade argument handling, and
the loud failure of the product path when no HIP device / library is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "yolo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tf2_yolo_amd import _lib
    names = _header_functions()
    assert len(names) >= 38
    lib = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU
    for n in names:
        assert hasattr(lib, n), f"libyolo_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    loaded = _lib.load()
    assert loaded.yolo_abi_version() >= 1
    assert loaded.yolo_device_available() in (0, 1)


def test_architecture_cross_checks():
    """SURVEY.md Appendix A / BASELINE.md section 3: parameter counts and forward FLOPs derived from the graph
    definitions equal the publicly known Darknet / keras-yolo3 figures."""
    from tf2_yolo_amd import graphs
    from tf2_yolo_amd.engine import conv_flops_per_image, count_params
    b3 = graphs.build_yolov3((416, 416, 3), 80)
    assert count_params(b3) == (61949149, 52608)            # 62 001 757 total
    assert round(conv_flops_per_image(b3) / 1e9, 3) == 65.864
    assert [(o.h, o.w, o.c) for o in b3.outputs] == [(13, 13, 255), (26, 26, 255), (52, 52, 255)]
    b4 = graphs.build_yolov4((608, 608, 3), 80)
    assert sum(count_params(b4)) == 64429405
    assert round(conv_flops_per_image(b4) / 1e9, 3) == 128.389
    b2 = graphs.build_yolov2((416, 416, 3), 20, [[1, 1]] * 5)
    assert round(conv_flops_per_image(b2) / 1e9, 3) == 29.715
    assert (b2.outputs[0].h, b2.outputs[0].w, b2.outputs[0].c) == (13, 13, 125)
    b1 = graphs.build_yolov1_5((224, 224, 3), 1, 2)
    assert round(conv_flops_per_image(b1) / 1e9, 3) == 10.147
    assert (b1.outputs[0].h, b1.outputs[0].w, b1.outputs[0].c) == (4, 4, 11)   # grid 4x4, not 224//64 = 3
    with pytest.raises(ValueError, match="multiple"):
        graphs.build_yolov3((416, 416, 3), 80, anchors=[[1, 1]] * 8)


def test_metric_string_grammar():
    from tf2_yolo_amd.facade import _metric_list, _parse_recall_threshold
    assert _parse_recall_threshold("obj+iou+recall0.5") == 0.5      # README.md:239
    assert _parse_recall_threshold("recall") == 0.5
    assert _parse_recall_threshold("mean_iou+recall0.6") == 0.6
    assert _parse_recall_threshold("recall0.75+obj_acc") == 0.75
    kinds = [m.kind for m in _metric_list("obj+iou+recall0.5", 3, (13, 13), 3, 80)]
    assert kinds == ["obj_acc", "mean_iou", "recall"]
    assert [m.kind for m in _metric_list("class_acc", 2, (13, 13), 5, 20)] == ["class_acc"]


def test_facade_surface_without_model():
    import yolov1_5
    import yolov2
    import yolov3
    import yolov4
    y3 = yolov3.Yolo((416, 416, 3), ["a"] * 20)
    assert (y3.grid_shape, y3.abox_num, y3.fpn_layers, y3.class_num, y3.model) == ((13, 13), 3, 3, 20, None)
    y4 = yolov4.Yolo((608, 608, 3), ["a"])
    assert y4.grid_shape == (19, 19) and y4.pan_layers == 3
    with pytest.raises(ValueError, match="haven't created a model"):
        y4.model
    with pytest.raises(ValueError, match="Can't set attribute"):
        y4.model = 1
    with pytest.raises(ValueError, match="create a model first"):
        y4.anchors
    with pytest.raises(ValueError, match="can't be empty"):
        y4.create_model()
    assert yolov2.Yolo().abox_num == 5 and yolov1_5.Yolo().grid_shape == (7, 7)
    assert yolov3.MetricKind.recall == "recall"
    with pytest.raises(ValueError, match="download"):
        y3.create_model()                                   # default pretrained_body="pascal_voc"
    with pytest.raises(NotImplementedError):
        y3.read_file_to_dataset("x", "y")


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tf2_yolo_amd import tools
    from tf2_yolo_amd._lib import YoloHipError
    with pytest.raises(YoloHipError):
        tools.decode(np.zeros((13, 13, 255), dtype=np.float32), class_num=80, version=3)
    import yolov3
    y = yolov3.Yolo((64, 64, 3), ["a"])
    with pytest.raises(YoloHipError):
        y.create_model(pretrained_body=None)
    with pytest.raises(ValueError, match="Invalid version"):
        tools.decode(np.zeros((13, 13, 255), dtype=np.float32), class_num=80, version=9)


# Public names of the reference's import surface, listed from its sources:
#   utils/tools.py (def / class at column 0), utils/measurement.py, yolovN/__init__.py:14-38 and the three
#   sub-packages' __init__.py (losses: cal_iou, wrap_yolo_loss; metrics: the four wrap_*; models: see below)
REFERENCE_SURFACE = {
    "utils.tools": ["read_img", "YoloDataSequence", "down2xlabel", "decode", "vis_img", "get_class_weight", "cal_iou",
                    "nms", "soft_nms", "create_score_mat", "array_to_json", "array_to_xml"],
    "utils.measurement": ["create_score_mat", "PRfunc", "PR_func"],
    "yolov3": ["Yolo", "MetricKind", "tools", "yolo_body", "tiny_yolo_body", "yolo_keras_app_body", "yolo_head",
               "wrap_yolo_loss", "wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov4": ["Yolo", "MetricKind", "tools", "yolo_body", "yolo_keras_app_body", "yolo_head", "wrap_yolo_loss",
               "wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov2": ["Yolo", "tools", "yolo_body", "yolo_head", "wrap_yolo_loss", "wrap_obj_acc", "wrap_mean_iou",
               "wrap_class_acc", "wrap_recall"],
    "yolov1_5": ["Yolo", "tools", "yolo_body", "yolo_head", "wrap_yolo_loss", "wrap_obj_acc", "wrap_mean_iou",
                 "wrap_class_acc", "wrap_recall"],
    "yolov3.losses": ["cal_iou", "wrap_yolo_loss"], "yolov4.losses": ["cal_iou", "wrap_yolo_loss"],
    "yolov2.losses": ["cal_iou", "wrap_yolo_loss"], "yolov1_5.losses": ["cal_iou", "wrap_yolo_loss"],
    "yolov3.metrics": ["wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov4.metrics": ["wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov2.metrics": ["wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov1_5.metrics": ["wrap_obj_acc", "wrap_mean_iou", "wrap_class_acc", "wrap_recall"],
    "yolov3.models": ["yolo_head", "yolo_body", "tiny_yolo_body", "yolo_keras_app_body", "darknet53"],
    "yolov4.models": ["yolo_head", "yolo_body", "yolo_keras_app_body", "csp_darknet53"],
    "yolov2.models": ["yolo_body", "yolo_head"],
    "yolov1_5.models": ["darknet", "yolo_body", "yolo_head"],
}


def test_reference_import_surface_resolves():
    """every public name a user of the reference imports exists under the same module path (row b)"""
    import importlib
    import inspect
    for mod, names in REFERENCE_SURFACE.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), f"{mod}.{n} is missing"
    # argument lists of the factories that differ between versions (yolov1_5/metrics/yolo_metrics.py:52: two arguments)
    import yolov1_5.metrics
    import yolov3.metrics
    assert list(inspect.signature(yolov1_5.metrics.wrap_class_acc).parameters) == ["grid_shape", "class_num"]
    assert list(inspect.signature(yolov3.metrics.wrap_class_acc).parameters) == ["grid_shape", "bbox_num", "class_num"]
    assert list(inspect.signature(yolov3.metrics.wrap_recall).parameters) == ["grid_shape", "bbox_num", "class_num",
                                                                              "iou_threshold"]
    import yolov4.losses
    assert list(inspect.signature(yolov4.losses.cal_iou).parameters) == ["xywh_true", "xywh_pred", "grid_shape",
                                                                         "return_ciou"]
    import utils.tools as T
    assert list(inspect.signature(T.cal_iou).parameters) == ["xywh_true", "xywh_pred", "mode"]
    assert list(inspect.signature(T.get_class_weight).parameters) == ["label_data", "method"]
    # out-of-scope entries fail with an explanation, not an AttributeError / ImportError
    for f, args in ((T.read_img, ("x.jpg",)), (T.vis_img, (None,)), (T.YoloDataSequence, ())):
        with pytest.raises(NotImplementedError, match="outside the accelerated hot path"):
            f(*args)
    import yolov3.models
    with pytest.raises(NotImplementedError, match="keras.applications"):
        yolov3.models.yolo_keras_app_body(None)
    with pytest.raises(ImportError, match="utils.measurement"):
        T.create_score_mat()


def test_symbolic_bodies_and_host_label_helpers():
    """yolo_body returns the symbolic graph (no device needed); utils.tools' host-side helpers equal the reference's
    outputs (tests/golden/tools_golden.npz)."""
    from yolov3.models import tiny_yolo_body, yolo_body
    import yolov1_5.models
    import yolov2.models
    import yolov4.models
    b = yolo_body((416, 416, 3))
    assert b.output_shape == [(None, 13, 13, 1024), (None, 26, 26, 512), (None, 52, 52, 256)]
    assert b.input_shape == (None, 416, 416, 3) and "block3_8_3x3_bn" in b.layer_names()
    assert tiny_yolo_body((416, 416, 3)).output_shape == [(None, 13, 13, 512), (None, 26, 26, 256)]
    assert yolov4.models.yolo_body((608, 608, 3)).output_shape == [(None, 19, 19, 1024), (None, 38, 38, 512),
                                                                   (None, 76, 76, 256)]
    assert yolov2.models.yolo_body((416, 416, 3)).output_shape == (None, 13, 13, 1024)
    assert yolov1_5.models.yolo_body((224, 224, 3)).output_shape == (None, 4, 4, 1024)
    with pytest.raises(ValueError, match="Invalid backbone"):
        yolov2.models.yolo_body(backbone="vgg")
    with pytest.raises(ValueError, match="download"):
        yolo_body(pretrained_weights="pascal_voc")
    from utils.tools import down2xlabel, get_class_weight
    sys_path_golden = os.path.join(ROOT, "tests", "golden")
    import sys
    sys.path.insert(0, sys_path_golden)
    try:
        import gen_inputs
    finally:
        sys.path.remove(sys_path_golden)
    g = np.load(os.path.join(sys_path_golden, "tools_golden.npz"))
    lab = gen_inputs.misc_inputs()["label52"]
    l26 = down2xlabel(lab)
    assert np.array_equal(l26, g["label26"]) and np.array_equal(down2xlabel(l26), g["label13"])
    assert np.array_equal(get_class_weight(lab[..., 4:5], "binary"), g["binary_weight52"])
    for m in ("alpha", "log", "effective"):
        assert np.array_equal(get_class_weight(lab[..., 5:], m), g[f"class_weight_{m}"])


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "tf2_yolo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_header_is_plain_c_and_a_c_host_links_the_library(tmp_path):
    """The boundary is a C-ABI: include/yolo_hip.h must compile as C99 (pedantic) and as C++, and a host written in C
    (examples/c_host/abi_check.c: no Python, no torch) must link libyolo_hip.so and get the ABI version, status codes and
    yolo_last_error() messages -- without a GPU (no compute call is made)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "yolo_hip.h")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = str(tmp_path / "abi_check")
    libdir = os.path.join(ROOT, "tf2_yolo_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_host", "abi_check.c"), "-o", exe, "-L" + libdir, "-lyolo_hip",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "abi 5" in r.stdout and "bad args" in r.stdout
