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
