"""The `onnx` stand-in (tests/onnx_shim) on its own: the constructors, array conversions and structural checks the reference's front-end
calls (onnx.py:10-29, 1138-1481) behave like the real package's on the cases the front-end produces. Skipped when the REAL `onnx`
package is importable (then nothing here is in use)."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import frontend_real as FR  # noqa: E402


@pytest.fixture(scope="module")
def onnx():
    if FR.ensure_onnx() != "shim":
        pytest.skip("the real onnx package is installed: the stand-in is not in use")
    import onnx as mod

    return mod


@pytest.mark.parametrize("dt", [np.float32, np.float16, np.float64, np.int64, np.int32, np.int8, np.uint8, np.bool_])
def test_array_round_trips(onnx, dt):
    from onnx import numpy_helper

    a = (np.arange(24).reshape(2, 3, 4) % 5).astype(dt)
    t = numpy_helper.from_array(a, "w")
    back = numpy_helper.to_array(onnx.load_model_from_string(onnx.helper.make_model(onnx.helper.make_graph([], "g", [], [], [t])).SerializeToString()).graph.initializer[0])
    assert back.dtype == a.dtype and back.shape == a.shape and np.array_equal(back, a) and t.name == "w"


def test_make_tensor_typed_fields_and_scalars(onnx):
    from onnx import TensorProto, helper, numpy_helper

    assert np.array_equal(numpy_helper.to_array(helper.make_tensor("s", TensorProto.INT64, [3], [4, -1, 7])), np.array([4, -1, 7], np.int64))
    assert np.allclose(numpy_helper.to_array(helper.make_tensor("f", TensorProto.FLOAT, [2, 2], [0.5, 1, 2, 3])), [[0.5, 1], [2, 3]])
    h = numpy_helper.to_array(helper.make_tensor("h", TensorProto.FLOAT16, [2], [1.5, -2.0]))
    assert h.dtype == np.float16 and np.array_equal(h, np.array([1.5, -2.0], np.float16))
    assert numpy_helper.to_array(helper.make_tensor("c", TensorProto.FLOAT, [], [0.125])).shape == ()


def test_make_node_attribute_kinds(onnx):
    from onnx import AttributeProto, helper

    n = helper.make_node("Conv", ["x", "w"], ["y"], "conv0", pads=[1, 1, 1, 1], group=1, alpha=0.5, mode="constant", flags=[0.5, 1.5])
    kinds = {a.name: a.type for a in n.attribute}
    assert kinds == {"pads": AttributeProto.INTS, "group": AttributeProto.INT, "alpha": AttributeProto.FLOAT, "mode": AttributeProto.STRING,
                     "flags": AttributeProto.FLOATS}
    by = {a.name: helper.get_attribute_value(a) for a in n.attribute}
    assert by["pads"] == [1, 1, 1, 1] and by["group"] == 1 and by["mode"] == b"constant" and by["flags"] == [0.5, 1.5]
    assert n.op_type == "Conv" and list(n.input) == ["x", "w"] and n.name == "conv0"


def test_value_info_and_checker(onnx):
    from onnx import TensorProto, checker, helper

    v = helper.make_tensor_value_info("x", TensorProto.FLOAT, [2, "N", None])
    dims = v.type.tensor_type.shape.dim
    assert dims[0].dim_value == 2 and dims[1].dim_param == "N" and not dims[2].HasField("dim_value") and not dims[2].HasField("dim_param")
    checker.check_value_info(v)
    good = helper.make_graph([helper.make_node("Relu", ["x"], ["y"])], "g", [v], [helper.make_tensor_value_info("y", TensorProto.FLOAT, [2])])
    checker.check_model(helper.make_model(good, opset_imports=[helper.make_opsetid("", 13)]))
    bad = helper.make_graph([helper.make_node("Relu", ["nowhere"], ["y"])], "g", [v], [helper.make_tensor_value_info("y", TensorProto.FLOAT, [2])])
    with pytest.raises(checker.ValidationError):
        checker.check_graph(bad)
    with pytest.raises(checker.ValidationError):
        checker.check_node(helper.make_node("", ["x"], ["y"]))


def test_messages_behave_like_protobuf(onnx):
    import copy

    from onnx import ModelProto, TensorProto, helper

    m = helper.make_model(helper.make_graph([helper.make_node("Relu", ["x"], ["y"])], "g",
                                            [helper.make_tensor_value_info("x", TensorProto.FLOAT, [1])],
                                            [helper.make_tensor_value_info("y", TensorProto.FLOAT, [1])]))
    m2 = copy.deepcopy(m)
    m2.graph.node[0].op_type = "Neg"
    assert m.graph.node[0].op_type == "Relu" and isinstance(m2, ModelProto) and m.HasField("graph")
    assert TensorProto.FLOAT == 1 and TensorProto.INT64 == 7 and TensorProto.FLOAT16 == 10 and TensorProto.BFLOAT16 == 16
    m3 = ModelProto()
    m3.ParseFromString(m2.SerializeToString())
    assert m3 == m2 and m3 != m
