"""The reference's Resize known-answer tests restated as a table (test/kernels/cuda/test_cuda_resize.cc): every row
cites the lines holding the input values, the sizes / scales / roi literals and the expected output."""
import numpy as np
from conftest import kat

RS = "test/kernels/cuda/test_cuda_resize.cc"
# in_shape, input line, ("sizes" | "scales", line), axes, mode, coord_mode, nearest_mode, roi line, expected line
CASES = [
    ((1, 1, 2, 4), 16, ("sizes", 17), None, "nearest", "half_pixel", "round_prefer_floor", None, 35),
    ((1, 1, 4, 4), 112, ("sizes", 113), None, "nearest", "half_pixel", "ceil", None, 134),
    ((1, 1, 4, 4), 149, ("sizes", 150), (3, 2), "nearest", "align_corners", "floor", None, 171),
    ((1, 1, 4, 4), 186, ("sizes", 187), None, "nearest", "asymmetric", "round_prefer_ceil", None, 209),
    ((1, 1, 2, 4), 223, ("scales", 224), None, "nearest", "half_pixel", "round_prefer_floor", None, 240),
    ((1, 1, 2, 2), 250, ("scales", 251), None, "nearest", "half_pixel", "round_prefer_floor", None, 268),
    ((1, 1, 2, 2), 279, ("scales", 280), (3, 2), "nearest", "half_pixel", "round_prefer_floor", None, 297),
    ((1, 1, 2, 4), 308, ("scales", 309), None, "linear", "half_pixel", "round_prefer_floor", None, 326),
    ((1, 1, 2, 4), 336, ("scales", 337), None, "linear", "align_corners", "round_prefer_floor", None, 355),
    ((1, 1, 2, 2), 365, ("scales", 366), None, "linear", "half_pixel", "round_prefer_floor", None, 384),
    ((1, 1, 2, 2), 395, ("scales", 396), None, "linear", "align_corners", "round_prefer_floor", None, 414),
    ((1, 1, 4, 4), 427, ("sizes", 428), None, "linear", "pytorch_half_pixel", "round_prefer_floor", None, 448),
    ((1, 1, 4, 4), 460, ("sizes", 461), None, "linear", "tf_crop_and_resize", "round_prefer_floor", 462, 484),
    ((1, 1, 4, 4), 497, ("sizes", 498), (3, 2), "linear", "tf_crop_and_resize", "round_prefer_floor", 499, 521),
    ((1, 1, 4, 4), 533, ("scales", 534), None, "cubic", "half_pixel", "round_prefer_floor", None, 553),
    ((1, 1, 4, 4), 565, ("scales", 566), None, "cubic", "align_corners", "round_prefer_floor", None, 585),
    ((1, 1, 4, 4), 597, ("scales", 598), None, "cubic", "half_pixel", "round_prefer_floor", None, 615),
    ((1, 1, 4, 4), 639, ("scales", 640), None, "cubic", "align_corners", "round_prefer_floor", None, 658),
    ((1, 1, 4, 4), 682, ("scales", 683), None, "cubic", "asymmetric", "round_prefer_floor", None, 701),
    ((1, 1, 4, 4), 721, ("sizes", 722), None, "cubic", "half_pixel", "round_prefer_floor", None, 746),
    ((1, 1, 4, 4), 759, ("sizes", 760), None, "cubic", "half_pixel", "round_prefer_floor", None, 778),
]


def materialise(case):
    """-> x, out_shape, scales (per dim), roi (2*ndim or None), expected: what ResizeObj derives from the test's
    tensors with the stretch policy (src/operators/resize.cc:75-123, 148-200)."""
    shape, xl, (kind, sl), axes, mode, coord, nearest, rl, el = case
    nd = len(shape)
    x = kat(RS, xl, "float").astype(np.float32).reshape(shape)
    axes = list(range(nd)) if axes is None else list(axes)
    scales = [1.0] * nd
    out = list(shape)
    vals = kat(RS, sl)
    for i, a in enumerate(axes):
        if kind == "sizes":
            out[a] = int(vals[i])
            scales[a] = float(np.float32(vals[i]) / np.float32(shape[a]))
        else:
            scales[a] = float(np.float32(vals[i]))
            out[a] = int(np.floor(np.float32(shape[a]) * np.float32(vals[i])))
    roi = None
    if rl is not None:
        r = kat(RS, rl, "float")
        roi = [0.0] * nd + [1.0] * nd
        for i, a in enumerate(axes):
            roi[a] = float(r[i])
            roi[a + nd] = float(r[i + len(axes)])
    return x, out, scales, roi, kat(RS, el, "float")
