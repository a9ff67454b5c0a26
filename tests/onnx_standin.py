"""Minimal stand-ins for the protobuf objects an ONNX importer touches (ModelProto.graph, NodeProto,
AttributeProto, TensorProto, ValueInfoProto) and for onnx.numpy_helper.to_array -- `onnx` itself is not
installed in the build image.  Shared by tools/capture_onnx_golden.py (which feeds these objects to the
REFERENCE's io.read_onnx through a fake `onnx` module) and tests/test_onnx_import.py (which feeds the
very same objects to planer_amd.onnx_import.graph_to_ir)."""
import types

import numpy as np

FLOAT, INT64 = 1, 7


class Attr:
    def __init__(self, name, i=0, f=0.0, s=b"", ints=(), t=None):
        self.name, self.i, self.f, self.s, self.ints, self.t = name, i, f, s, list(ints), t


class Tensor:
    def __init__(self, name, array):
        self.name, self.array = name, np.asarray(array)
        self.dims = list(self.array.shape)
        self.data_type = {"float32": FLOAT, "int64": INT64}.get(str(self.array.dtype), FLOAT)


class Node:
    def __init__(self, op_type, inputs, outputs, name=None, attrs=()):
        self.op_type, self.input, self.output = op_type, list(inputs), list(outputs)
        self.name = name or (op_type.lower() + "_" + outputs[0])
        self.attribute = list(attrs)


def to_array(t):
    return t.array


def model(inputs, outputs, inits, nodes):
    g = types.SimpleNamespace(input=[types.SimpleNamespace(name=n) for n in inputs],
                              output=[types.SimpleNamespace(name=n) for n in outputs],
                              initializer=list(inits), node=list(nodes))
    return types.SimpleNamespace(graph=g)


def mini_resnet():
    """Executable: conv(+attrs) - bn - relu - maxpool - conv(no attrs) - bn - add - relu - gap - flatten - gemm."""
    rng = np.random.default_rng(11)
    f32 = lambda *s, scale=1.0: (rng.standard_normal(s) * scale).astype(np.float32)
    inits = [Tensor("w1", f32(8, 3, 3, 3, scale=0.2)), Tensor("b1", f32(8)),
             Tensor("g1", rng.uniform(0.5, 1.5, 8).astype(np.float32)), Tensor("be1", f32(8, scale=0.1)),
             Tensor("m1", f32(8, scale=0.1)), Tensor("v1", rng.uniform(0.5, 1.5, 8).astype(np.float32)),
             Tensor("w2", f32(8, 8, 1, 1, scale=0.3)),
             Tensor("g2", rng.uniform(0.5, 1.5, 8).astype(np.float32)), Tensor("be2", f32(8, scale=0.1)),
             Tensor("m2", f32(8, scale=0.1)), Tensor("v2", rng.uniform(0.5, 1.5, 8).astype(np.float32)),
             Tensor("fcw", f32(5, 8, scale=0.3)), Tensor("fcb", f32(5))]
    nodes = [
        Node("Conv", ["x", "w1", "b1"], ["c1"], attrs=[Attr("dilations", ints=[1, 1]), Attr("group", i=1),
                                                        Attr("kernel_shape", ints=[3, 3]), Attr("pads", ints=[1, 1, 1, 1]),
                                                        Attr("strides", ints=[1, 1])]),
        Node("BatchNormalization", ["c1", "g1", "be1", "m1", "v1"], ["n1"], attrs=[Attr("epsilon", f=1e-3)]),
        Node("Relu", ["n1"], ["r1"]),
        Node("MaxPool", ["r1"], ["p1"], attrs=[Attr("kernel_shape", ints=[2, 2]), Attr("pads", ints=[0, 0, 0, 0]),
                                               Attr("strides", ints=[2, 2])]),
        Node("Conv", ["p1", "w2"], ["c2"], attrs=[Attr("pads", ints=[0, 0, 0, 0]), Attr("strides", ints=[1, 1]),
                                                  Attr("dilations", ints=[1, 1])]),
        Node("BatchNormalization", ["c2", "g2", "be2", "m2", "v2"], ["n2"]),
        Node("Add", ["n2", "p1"], ["a1"]),
        Node("Relu", ["a1"], ["r2"]),
        Node("GlobalAveragePool", ["r2"], ["gp"]),
        Node("Flatten", ["gp"], ["fl"], attrs=[Attr("axis", i=1)]),
        Node("Gemm", ["fl", "fcw", "fcb"], ["y"], attrs=[Attr("alpha", f=1.0), Attr("beta", f=1.0), Attr("transB", i=1)]),
    ]
    return model(["x"], ["y"], inits, nodes)


def every_op():
    """Not executable: one node per op type of the importer's table, with the attribute variants that
    matter (absent / present / zero-valued / first-attribute reads)."""
    f = np.float32
    inits = [Tensor("W", np.ones((4, 2, 3, 3), f)), Tensor("Wt", np.ones((2, 4, 2, 2), f)), Tensor("FC", np.ones((7, 3), f)),
             Tensor("scalar", np.array(2.5, f)), Tensor("idx", np.array([0, 2], np.int64)),
             Tensor("g", np.ones(4, f)), Tensor("b", np.zeros(4, f)), Tensor("m", np.zeros(4, f)), Tensor("v", np.ones(4, f))]
    n = []
    add = lambda *a, **k: n.append(Node(*a, **k))
    add("Conv", ["x", "W"], ["t0"])                                                  # no attributes at all
    add("Conv", ["x", "W"], ["t1"], attrs=[Attr("group", i=2), Attr("strides", ints=[2, 2])])
    add("ConvTranspose", ["t1", "Wt"], ["t2"], attrs=[Attr("strides", ints=[2, 2]), Attr("output_padding", ints=[1, 1])])
    add("BatchNormalization", ["t0", "g", "b", "m", "v"], ["t3"], attrs=[Attr("epsilon", f=0.5)])
    add("Gemm", ["t3", "FC", "b"], ["t4"])
    add("MaxPool", ["t0"], ["t5"], attrs=[Attr("kernel_shape", ints=[3, 3]), Attr("strides", ints=[2, 2])])    # pads absent
    add("AveragePool", ["t0"], ["t6"], attrs=[Attr("kernel_shape", ints=[2, 2]), Attr("pads", ints=[0, 0, 1, 1]),
                                              Attr("strides", ints=[2, 2])])
    add("GlobalAveragePool", ["t0"], ["t7"])
    add("Upsample", ["t0", "scalar"], ["t8"], attrs=[Attr("mode", s=b"nearest")])
    add("Upsample", ["t0", "scalar"], ["t8b"])
    add("Resize", ["t0", "scalar", "scalar"], ["t9"], attrs=[Attr("coordinate_transformation_mode", s=b"asymmetric"),
                                                             Attr("mode", s=b"nearest"), Attr("nearest_mode", s=b"floor")])
    add("Flatten", ["t0"], ["t10"])
    add("Unsqueeze", ["t0"], ["t11"], attrs=[Attr("axes", ints=[0, 2])])
    add("Unsqueeze", ["t0", "idx"], ["t12"])
    add("Squeeze", ["t0"], ["t13"], attrs=[Attr("axes", ints=[1])])
    add("Squeeze", ["t0"], ["t14"])
    add("Relu", ["t0"], ["t15"])
    add("LeakyRelu", ["t0"], ["t16"], attrs=[Attr("alpha", f=0.125)])
    add("HardSigmoid", ["t0"], ["t17"], attrs=[Attr("alpha", f=0.25)])
    add("HardSigmoid", ["t0"], ["t18"], attrs=[Attr("beta", f=0.75), Attr("alpha", f=0.5)])
    for k, op in enumerate(["Add", "Sub", "Div", "Mul", "Pow", "MatMul", "Tile", "Greater", "GreaterOrEqual", "Equal",
                            "Where", "Range", "ScatterND", "Expand", "Slice", "Reshape"]):
        add(op, ["t0", "t1"] + (["t3"] if op in ("Where", "Range", "ScatterND", "Slice") else []), ["b%d" % k])
    for k, op in enumerate(["Identity", "Sigmoid", "Tanh", "Exp", "Log", "Sqrt", "Erf", "Reciprocal", "Shape", "NonZero"]):
        add(op, ["t0"], ["u%d" % k])
    add("Constant", [], ["c0"], attrs=[Attr("value", t=Tensor("", np.array(3, np.int64)))])           # 0-d
    add("Constant", [], ["c1"], attrs=[Attr("value", t=Tensor("", np.array([1.5, 2.5], f)))])
    add("ConstantOfShape", ["idx"], ["c2"], attrs=[Attr("value", t=Tensor("", np.array([7], np.int64)))])
    add("ConstantOfShape", ["idx"], ["c3"], attrs=[Attr("value", t=Tensor("", np.array([0.0, 1.0], f)))])
    for k, op in enumerate(["ReduceSum", "ReduceMean", "ReduceMax", "ReduceMin"]):
        add(op, ["t0"], ["r%d" % k], attrs=[Attr("axes", ints=[2, 3]), Attr("keepdims", i=k % 2)] if k < 3 else [])
    add("Concat", ["t0", "t1"], ["k0"], attrs=[Attr("axis", i=1)])
    add("Pad", ["t0", "idx"], ["k1"], attrs=[Attr("mode", s=b"constant")])
    add("Pad", ["t0"], ["k2"], attrs=[Attr("mode", s=b"reflect"), Attr("constant_value", f=0.5)])
    add("LSTM", ["t0", "W", "W", "b"], ["k3", "k3h", "k3c"], attrs=[Attr("hidden_size", i=16), Attr("direction", s=b"bidirectional")])
    add("Gather", ["t0", "idx"], ["k4"])
    add("Gather", ["t0", "idx"], ["k5"], attrs=[Attr("axis", i=2)])
    add("Transpose", ["t0"], ["k6"], attrs=[Attr("perm", ints=[0, 2, 3, 1])])
    add("Transpose", ["t0"], ["k7"])
    add("LogSoftmax", ["t0"], ["k8"], attrs=[Attr("axis", i=-1)])
    add("Softmax", ["t0"], ["k9"], attrs=[Attr("axis", i=1)])
    add("TopK", ["t0", "idx"], ["k10", "k10i"], attrs=[Attr("axis", i=-1), Attr("largest", i=1), Attr("sorted", i=1)])
    add("TopK", ["t0", "idx"], ["k11", "k11i"])
    add("Split", ["t0"], ["s0", "s1"], attrs=[Attr("axis", i=1), Attr("split", ints=[1, 3])])
    add("Split", ["t0"], ["s2", "s3"])
    add("Cast", ["t0"], ["k12"], attrs=[Attr("to", i=7)])
    add("InstanceNormalization", ["t0", "g", "b"], ["k13"], attrs=[Attr("epsilon", f=1e-3)])
    add("Clip", ["t0"], ["k14"], attrs=[Attr("min", f=0.0), Attr("max", f=6.0)])        # min 0.0 is dropped by the truth test
    add("Clip", ["t0"], ["k15"], attrs=[Attr("min", f=-1.0)])
    add("Clip", ["t0", "scalar", "scalar"], ["k16"])
    return model(["x", "W"], ["k16", "t4", "s3"], inits, n)


def unknown_op():
    return model(["x"], ["y"], [], [Node("Relu", ["x"], ["a"]), Node("Einsum", ["a"], ["y"], attrs=[Attr("equation", s=b"ij->ji")])])


MODELS = {"mini_resnet": mini_resnet, "every_op": every_op, "unknown_op": unknown_op}
