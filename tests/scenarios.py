"""Geometry scenarios shared by the golden-vector generator and the parity tests
(identity, scale, partial cover with border+fill, rotation,
opacity, odd sizes)."""
import util

LAYER_KERNELS_REF = ["img_nv12_nv12", "img_y420p_nv12", "img_y420p_y420p", "img_bgra_y420p",
                     "img_rgba_y420p", "img_bgra_nv12", "img_rgba_nv12"]
LAYER_KERNELS_OWN = ["img_nv12_bgra", "img_y420p_bgra", "img_bgra_bgra_tx", "img_rgba_bgra_tx"]
# integer BT.601/709 RGB -> YUV; not part of the committed golden set (tests/golden/gen_golden.py)
LAYER_KERNELS_INT = ["img_bgra_nv12_int", "img_rgba_nv12_int", "img_bgra_y420p_int", "img_rgba_y420p_int"]
CLEAR_KERNELS = ["img_clear_nv12", "img_clear_y420p", "img_clear_bgra"]

# name -> (canvas w, h, input w, h, make_uniforms kwargs)
SCENARIOS = {
    "identity":   (64, 36, 64, 36, dict()),
    "downscale":  (64, 36, 96, 54, dict()),
    "upscale":    (64, 36, 24, 14, dict(opacity=0.75)),
    "rect_fill":  (64, 36, 40, 30, dict(rect=(10, 6, 30, 20), border=(3, 2, 4, 1), fill=(0.2, 0.7, 0.4, 0.8), opacity=0.6)),
    "rotated":    (64, 36, 40, 30, dict(rect=(20, 4, 30, 24), rotation=0.5235987755982988, border=(2, 2, 2, 2), fill=(1, 0, 0, 1), opacity=0.9)),
    "offscreen":  (33, 17, 21, 9, dict(rect=(-5, -3, 30, 18), opacity=0.5, tex=(0.1, 0.05, 0.8, 0.9), fill=(0.1, 0.2, 0.3, 0.5))),
    "letterbox":  (64, 36, 32, 32, dict(rect=(8, 2, 48, 32), tex=(0.25, 0.0, 0.5, 1.0), fill=(0.9, 0.9, 0.1, 1.0))),
    "odd_7x5":    (7, 5, 7, 5, dict()),
    "tiny_2x2":   (2, 2, 2, 2, dict(opacity=0.3)),
}


def uniforms_for(name):
    cw, ch, iw, ih, kw = SCENARIOS[name]
    return util.make_uniforms((cw, ch), in_size=(iw, ih), **kw)
