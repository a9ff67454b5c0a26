"""ONNX graph -> planer IR: a restatement of the reference's importer,
io.read_onnx (io.py:53-287) and io.onnx2pla (io.py:289-299).

The importer is format plumbing, not compute: it turns a `GraphProto` into the
`{input, inits, layers, flow}` json + the uint8 weight blob that `Net.load_json`
/ `Net.load_weights` take.  `graph_to_ir` only touches the handful of protobuf
attributes listed below, so it is testable without the `onnx` package (absent
from this image) against stand-in objects; `read_onnx(path)` needs `onnx`.

Protobuf surface used (same as the reference): graph.input[i].name,
graph.output[i].name, graph.initializer (-> numpy via `to_array`), and per node
`.name .op_type .input .output .attribute[j].{name,i,f,s,ints,t}`.

Reference behaviours kept on purpose (they define what an imported file looks like):
  * BatchNormalization is folded at import: K = gamma/sqrt(var + 1e-5), B = beta - gamma*mean/
    sqrt(var + 1e-5) as two new (1,C,1,1) inits `<gamma>_invK`, `<gamma>_invB`; the epsilon is
    the constant 1e-5, the node's own `epsilon` attribute is ignored (io.py:76-91);
  * `Constant` nodes become inits named after their output and leave no layer (io.py:156-165);
  * a `return` layer over the graph outputs is appended, writing `plrst` (io.py:284-285);
  * 0-d tensors are stored as one element (io.py:62-63);
  * several ops read "the first attribute" rather than an attribute by name (LeakyRelu alpha,
    Concat / Softmax / LogSoftmax axis, Cast to, LSTM hidden_size, InstanceNormalization epsilon);
  * `Clip` drops a min / max of exactly 0 (truth test, io.py:273-278); `Reciprocal` is emitted as
    kind `erf` (io.py:269-270 -- a slip in the reference; kept so files match);
  * an unknown op prints `lost layer: <op>` and yields ('lost', node) (io.py:280-282).
"""
import json
import os
import zipfile

import numpy

# TensorProto.DataType -> numpy dtype name (io.py:36-37)
ONNX_TYPES = [None, "float32", "uint8", "int8", "uint16", "int16", "int32", "int64", "str", "bool", "float16",
              "float64", "uint32", "uint64", "complex64", "complex128"]


def _attr(attrs, name, field=None, into=None):
    """The attribute called `name` (last one wins), or its `field`; ints -> list, s -> str.  With
    `into`, a present attribute is also recorded there under its name (io.py:39-50)."""
    hit = None
    for a in attrs:
        if a.name == name:
            hit = a
    if field is None or hit is None:
        return hit
    val = getattr(hit, field)
    if field == "ints":
        val = list(val)
    elif field == "s":
        val = val.decode()
    if into is not None:
        into[name] = val
    return val


def _named(*spec):
    """para = the listed attributes that are present, by name: spec = (name, field) pairs."""
    def build(node, ctx):
        para = {}
        for name, field in spec:
            _attr(node.attribute, name, field, para)
        return para
    return build


def _always(*spec):
    """para = every listed key, None when the attribute is absent: spec = (key, attribute, field)."""
    def build(node, ctx):
        return {key: _attr(node.attribute, name, field) for key, name, field in spec}
    return build


def _first(key, field, convert=None):
    """para = {key: first attribute's `field`} (the reference does not look the name up)."""
    def build(node, ctx):
        v = getattr(node.attribute[0], field)
        return {key: convert(v) if convert else v}
    return build


def _conv(node, ctx):
    a = node.attribute
    return {"group": _attr(a, "group", "i") or 1, "strides": _attr(a, "strides", "ints"),
            "dilations": _attr(a, "dilations", "ints"), "pads": _attr(a, "pads", "ints")}


def _gemm(node, ctx):
    return {"shp": list(ctx["values"][node.input[1]][1][::-1])}       # the weight must be an initializer


def _axes_if_present(node, ctx):
    axes = _attr(node.attribute, "axes", "ints")
    return {} if axes is None else {"axes": axes}


def _split(node, ctx):
    para = {"axis": _attr(node.attribute, "axis", "i")}
    split = _attr(node.attribute, "split", "ints")
    if split is not None:
        para["split"] = split
    return para


def _clip(node, ctx):
    para = {}
    lo, hi = _attr(node.attribute, "min", "f"), _attr(node.attribute, "max", "f")
    if lo:
        para["min"] = lo
    if hi:
        para["max"] = hi
    return para


def _constant_of_shape(node, ctx):
    v = ctx["to_array"](node.attribute[0].t)
    vals = v.tolist()
    return {"value": vals[0] if len(vals) == 1 else 0, "dtype": str(v.dtype)}


_POOL = _always(("w", "kernel_shape", "ints"), ("pads", "pads", "ints"), ("strides", "strides", "ints"))
_REDUCE = _named(("axes", "ints"), ("keepdims", "i"))
_NONE = None

# op_type -> (planer layer kind, how its json parameters are made)
OP_TABLE = {
    "Conv": ("conv", _conv),
    "ConvTranspose": ("convtranspose", _named(("group", "i"), ("dilations", "ints"), ("pads", "ints"), ("strides", "ints"),
                                              ("output_padding", "ints"))),
    "Gemm": ("dense", _gemm),
    "MaxPool": ("maxpool", _POOL),
    "AveragePool": ("averagepool", _POOL),
    "GlobalAveragePool": ("gap", _NONE),
    "Upsample": ("upsample", _always(("mode", "mode", "s"))),
    "Resize": ("resize", _always(("mode", "mode", "s"), ("nearest_mode", "nearest_mode", "s"),
                                 ("coordinate_transformation_mode", "coordinate_transformation_mode", "s"))),
    "Flatten": ("flatten", _NONE), "Unsqueeze": ("unsqueeze", _axes_if_present), "Squeeze": ("squeeze", _axes_if_present),
    "Relu": ("relu", _NONE), "LeakyRelu": ("leakyrelu", _first("alpha", "f")),
    "HardSigmoid": ("hardsigmoid", _named(("alpha", "f"), ("beta", "f"))),
    "Sigmoid": ("sigmoid", _NONE), "Tanh": ("tanh", _NONE), "Exp": ("exp", _NONE), "Log": ("log", _NONE),
    "Sqrt": ("sqrt", _NONE), "Erf": ("erf", _NONE), "Reciprocal": ("erf", _NONE),
    "Add": ("add", _NONE), "Sub": ("sub", _NONE), "Mul": ("mul", _NONE), "Div": ("div", _NONE), "Pow": ("pow", _NONE),
    "MatMul": ("matmul", _NONE), "Tile": ("tile", _NONE), "Identity": ("identity", _NONE),
    "ReduceSum": ("reducesum", _REDUCE), "ReduceMean": ("reducemean", _REDUCE), "ReduceMax": ("reducemax", _REDUCE),
    "ReduceMin": ("reducemin", _REDUCE),
    "Concat": ("concat", _first("axis", "i")),
    "Pad": ("pad", _named(("mode", "s"), ("constant_value", "f"))),
    "LSTM": ("lstm", lambda node, ctx: dict(_first("hidden_size", "i")(node, ctx), **_named(("direction", "s"))(node, ctx))),
    "Shape": ("shape", _NONE),
    "Gather": ("gather", lambda node, ctx: {"axis": _attr(node.attribute, "axis", "i") or 0}),
    "Reshape": ("reshape", _NONE),
    "Transpose": ("transpose", _always(("axis", "perm", "ints"))),
    "LogSoftmax": ("logsoftmax", _first("axis", "i")), "Softmax": ("softmax", _first("axis", "i")),
    "ConstantOfShape": ("constantofshape", _constant_of_shape),
    "Greater": ("greater", _NONE), "GreaterOrEqual": ("greaterorequal", _NONE), "Equal": ("equal", _NONE),
    "NonZero": ("nonzero", _NONE), "Where": ("where", _NONE), "Range": ("range", _NONE),
    "TopK": ("topk", _named(("axis", "i"), ("largest", "i"), ("sorted", "i"))),
    "Split": ("split", _split),
    "Slice": ("slice", _NONE), "Expand": ("expand", _NONE), "ScatterND": ("scatternd", _NONE),
    "Cast": ("cast", _first("dtype", "i", lambda t: ONNX_TYPES[t])),
    "InstanceNormalization": ("instancenormalization", _first("epsilon", "f")),
    "Clip": ("clip", _clip),
}


def graph_to_ir(graph, to_array):
    """-> ({'input', 'inits', 'layers', 'flow'}, uint8 blob), or ('lost', node) for an unknown op."""
    inputs = [i.name for i in graph.input]
    layers, inits, tensors, flows = [], [], [], []
    values = {}                       # tensor name -> (index into tensors, shape)

    def add_init(name, arr):
        values[name] = (len(tensors), arr.shape)
        inits.append([name, arr.shape, str(arr.dtype)])
        tensors.append(numpy.array([arr]) if arr.ndim == 0 else arr)

    for t in graph.initializer:
        add_init(t.name, to_array(t))
    ctx = {"values": values, "to_array": to_array}
    for node in graph.node:
        src, dst = list(node.input), list(node.output)
        src = src[0] if len(src) == 1 else src
        dst = dst[0] if len(dst) == 1 else dst
        op = node.op_type
        if op == "Constant":                       # io.py:156-165: an init named after the output, no layer
            add_init(dst, to_array(node.attribute[0].t))
            continue
        if op == "BatchNormalization":             # io.py:76-91
            gamma, beta, mean, var = [tensors[values[src[j]][0]] for j in (1, 2, 3, 4)]
            inv = 1 / numpy.sqrt(var + 1e-5)
            shift = -gamma * mean * inv + beta
            scale = gamma * inv
            scale.shape = shift.shape = (1, -1, 1, 1)
            kname, bname = src[1] + "_invK", src[1] + "_invB"
            add_init(kname, scale)
            add_init(bname, shift)
            flows.append([[src[0], kname, bname], [node.name], dst])
            layers.append([node.name, "batchnorm", {}])
            continue
        if op not in OP_TABLE:
            print("lost layer:", op)
            return "lost", node
        kind, make = OP_TABLE[op]
        flows.append([src, [node.name], dst])
        layers.append([node.name, kind, make(node, ctx) if make else {}])
    layers.append(["return", "return", {}])
    flows.append([[o.name for o in graph.output], ["return"], "plrst"])
    blob = numpy.hstack([t.view(dtype=numpy.uint8).ravel() for t in tensors]) if tensors else numpy.zeros(0, numpy.uint8)
    return {"input": inputs, "inits": inits, "layers": layers, "flow": flows}, blob


def read_onnx(path):
    """io.read_onnx: needs the `onnx` package."""
    try:
        import onnx
        import onnx.numpy_helper
    except ImportError as e:
        raise ImportError("reading %s needs the `onnx` package (reference io.py:53-54); convert the model to "
                          ".pla / .json+.npy where onnx is installed (onnx2pla)" % path) from e
    return graph_to_ir(onnx.load(path).graph, onnx.numpy_helper.to_array)


def _jsonable(graph):
    return json.loads(json.dumps(graph, default=lambda o: o.tolist() if hasattr(o, "tolist") else list(o)))


def onnx2pla(path, zip=True):
    """io.onnx2pla (io.py:289-299): `<model>.onnx` -> `<model>.pla` (or `.json` + `.npy`)."""
    graph, blob = read_onnx(path)
    base = path[:-5] if path.endswith(".onnx") else path
    numpy.save(base + ".npy", blob)
    with open(base + ".json", "w") as f:
        json.dump(_jsonable(graph), f)
    if zip:
        with zipfile.ZipFile(base + ".pla", "w") as z:
            z.write(base + ".json", os.path.split(base)[1] + ".json")
            z.write(base + ".npy", os.path.split(base)[1] + ".npy")
        os.remove(base + ".json")
        os.remove(base + ".npy")
