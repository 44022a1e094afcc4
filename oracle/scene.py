"""Oracle restatement (numpy) of view_to_packed_data / pack_rgba, brush-dataset/src/scene.rs:97-136.
TEST INFRASTRUCTURE ONLY."""
import numpy as np


def view_to_packed_data(img_u8, transparent_alpha=True):
    """img_u8 [H,W,3] or [H,W,4] uint8 -> (packed [H,W] uint32 little-endian r g b a, has_alpha).
    RGB views get a = 255; RGBA views with AlphaMode::Transparent are premultiplied in byte space,
    mul(c) = (c * a + 127) / 255 (integer); AlphaMode::Masked keeps the colours."""
    a = np.asarray(img_u8)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] in (3, 4)
    has_alpha = a.shape[2] == 4
    v = a.astype(np.uint32)
    if not has_alpha:
        return (v[..., 0] | (v[..., 1] << 8) | (v[..., 2] << 16) | np.uint32(255 << 24)).astype(np.uint32), False
    al = v[..., 3]
    if transparent_alpha:
        rgb = [(v[..., c] * al + 127) // 255 for c in range(3)]
    else:
        rgb = [v[..., c] for c in range(3)]
    return (rgb[0] | (rgb[1] << 8) | (rgb[2] << 16) | (al << 24)).astype(np.uint32), True
