"""Copies the reference's golden forward tensors into tests/golden/ as .npz.

Source: /root/reference/crates/brush-bench-test/test_cases/{tiny,basic}_case.safetensors
(gsplat-CUDA renders used by crates/brush-bench-test/src/reference.rs:80-151).
mix_case.safetensors is a missing LFS blob in the reference snapshot.
Run in the build container only (the GPU box has no /root/reference).
"""
import os
import numpy as np
from safetensors.numpy import load_file

SRC = "/root/reference/crates/brush-bench-test/test_cases"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(DST, exist_ok=True)
for name in ("tiny_case", "basic_case"):
    d = load_file(os.path.join(SRC, name + ".safetensors"))
    np.savez_compressed(os.path.join(DST, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items()})

# ---- the reference's only dataset fixture (apps/brush-c/tests/data/test_dataset, driven by
# apps/brush-c/tests/integration.rs:40-183): a 100-point trimesh cloud with uchar colours, one 50x50 RGBA view and its
# nerfstudio transforms.json.  Kept as bytes / decoded pixels / parsed numbers in ONE npz so the GPU box (no /root/reference,
# no PNG decoder needed) can run it end to end: tests/test_gpu_reference_fixture.py.
import json
from PIL import Image
FIX = "/root/reference/apps/brush-c/tests/data/test_dataset"
tf = json.load(open(os.path.join(FIX, "transforms.json")))
fr = tf["frames"][0]
rgba = np.array(Image.open(os.path.join(FIX, fr["file_path"])))
assert rgba.dtype == np.uint8 and rgba.shape == (50, 50, 4)
np.savez_compressed(
    os.path.join(DST, "test_dataset.npz"),
    init_ply=np.frombuffer(open(os.path.join(FIX, "init.ply"), "rb").read(), np.uint8),
    r_0_rgba=rgba,
    transform_matrix=np.array(fr["transform_matrix"], np.float32),
    intrinsics=np.array([fr["fl_x"], fr["fl_y"], fr["cx"], fr["cy"], fr["w"], fr["h"], fr["camera_angle_x"], fr["camera_angle_y"]], np.float64))
print("test_dataset", rgba.shape, fr["file_path"])
