#!/usr/bin/env python
"""Turn the reference's archived TF1 graphs (test_v1/model/*.meta) into data fixtures.

Runs ONLY in the build container (needs /root/reference).  A .meta file is a serialized
``MetaGraphDef``; field 2 is the ``GraphDef`` whose ``NodeDef`` entries (name, op, inputs, attrs incl. the
``Const`` tensors) describe the reference's own forward graph AND the ``gradients/...`` subgraph TensorFlow's
autodiff generated for it, the Adam update ops and every constant the scripts baked in (epsilon 1e-9, sqrt 2, clip 8,
leaky alpha, Adam betas, the decay schedule, the regulariser scales, the glorot limits, conv attributes, transpose
permutations, reshape targets).  No TensorFlow or protoc is needed: a small varint / length-delimited walker suffices.

Output: tests/golden/v1_graph/<checkpoint>.json.gz -- one JSON manifest per archived graph:
    {"source": ..., "nodes": [{"name", "op", "inputs": [...], "attr": {key: [kind, value]}}, ...]}
``save/*`` nodes and the bookkeeping attrs ``_class`` / ``_output_shapes`` are dropped; tensors larger than 4096
elements (none in these graphs) would be stored without values.  These manifests are DATA (structure + constants of a
graph the reference ships); oracle/tf_graph.py evaluates them with NumPy and tests/test_graph_golden.py uses that to pin
oracle/dccn_oracle.py -- forward and hand-derived backward -- against TensorFlow's own graph.

    python tests/golden/make_graph_golden.py
"""
import glob
import gzip
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _proto_fields, _varint  # noqa: E402

REF_MODELS = "/root/reference/test_v1/model"
DT = {1: "float32", 2: "float64", 3: "int32", 7: "string", 8: "complex64", 9: "int64", 10: "bool", 19: "float16"}
NP = {"float32": "<f4", "float64": "<f8", "int32": "<i4", "int64": "<i8", "bool": "?", "float16": "<f2"}


def _sint(x):
    x = int(x)
    return x - (1 << 64) if x >= (1 << 63) else x


def _packed_varints(wt, v):
    if wt == 0:
        return [_sint(v)]
    out, p, b = [], 0, bytes(v)
    while p < len(b):
        x, p = _varint(b, p)
        out.append(_sint(x))
    return out


def parse_shape(buf):
    dims, unknown = [], False
    for f, _, v in _proto_fields(buf):
        if f == 2:
            sz = 0
            for f3, _, d in _proto_fields(v):
                if f3 == 1:
                    sz = _sint(d)
            dims.append(sz)
        elif f == 3:
            unknown = bool(v)
    return None if unknown else dims


def parse_tensor(buf):
    dtype, shape, content = None, [], None
    vals = []
    for f, wt, v in _proto_fields(buf):
        if f == 1:
            dtype = DT.get(int(v), int(v))
        elif f == 2:
            shape = parse_shape(v)
        elif f == 4:
            content = bytes(v)
        elif f == 5:                                   # float_val
            vals += [struct.unpack("<f", v)[0]] if wt == 5 else np.frombuffer(bytes(v), "<f4").tolist()
        elif f == 6:                                   # double_val
            vals += [struct.unpack("<d", v)[0]] if wt == 1 else np.frombuffer(bytes(v), "<f8").tolist()
        elif f in (7, 10):                             # int_val / int64_val
            vals += _packed_varints(wt, v)
        elif f == 11:                                  # bool_val
            vals += [bool(x) for x in _packed_varints(wt, v)]
        elif f == 13:                                  # half_val (stored as uint16 bit patterns)
            vals += [float(np.array(x, dtype=np.uint16).view(np.float16)) for x in _packed_varints(wt, v)]
        elif f == 8:
            vals.append(bytes(v).decode("latin1"))
    n = int(np.prod(shape)) if shape else 1
    if content is not None and dtype in NP:
        vals = np.frombuffer(content, NP[dtype]).tolist()
    if len(vals) == 1 and n > 1:
        vals = vals * n                                # TensorProto splat
    if not vals and n >= 1 and dtype in NP:
        vals = [0] * n                                 # proto3 default
    return dict(dtype=dtype, shape=shape, value=vals if len(vals) <= 4096 else None)


def parse_attr(buf):
    for f, wt, v in _proto_fields(buf):
        if f == 2:
            return ["s", bytes(v).decode("latin1")]
        if f == 3:
            return ["i", _sint(v)]
        if f == 4:
            return ["f", struct.unpack("<f", v)[0]]
        if f == 5:
            return ["b", bool(v)]
        if f == 6:
            return ["type", DT.get(int(v), int(v))]
        if f == 7:
            return ["shape", parse_shape(v)]
        if f == 8:
            return ["tensor", parse_tensor(v)]
        if f == 1:
            out = {}
            for f2, wt2, v2 in _proto_fields(v):
                if f2 == 2:
                    out.setdefault("s", []).append(bytes(v2).decode("latin1"))
                elif f2 == 3:
                    out.setdefault("i", []).extend(_packed_varints(wt2, v2))
                elif f2 == 4:
                    out.setdefault("f", []).extend(
                        [struct.unpack("<f", v2)[0]] if wt2 == 5 else np.frombuffer(bytes(v2), "<f4").tolist())
                elif f2 == 5:
                    out.setdefault("b", []).extend(bool(x) for x in _packed_varints(wt2, v2))
                elif f2 == 6:
                    out.setdefault("type", []).extend(DT.get(x, x) for x in _packed_varints(wt2, v2))
                elif f2 == 7:
                    out.setdefault("shape", []).append(parse_shape(v2))
            return ["list", out]
    return ["none", None]


def parse_meta(path):
    buf = open(path, "rb").read()
    nodes = []
    for f, _, v in _proto_fields(buf):
        if f != 2:                                     # MetaGraphDef.graph_def
            continue
        for f2, _, v2 in _proto_fields(v):
            if f2 != 1:                                # GraphDef.node
                continue
            nd = dict(name=None, op=None, inputs=[], attr={})
            for f3, _, v3 in _proto_fields(v2):
                if f3 == 1:
                    nd["name"] = bytes(v3).decode()
                elif f3 == 2:
                    nd["op"] = bytes(v3).decode()
                elif f3 == 3:
                    nd["inputs"].append(bytes(v3).decode())
                elif f3 == 5:
                    key = val = None
                    for f4, _, v4 in _proto_fields(v3):
                        if f4 == 1:
                            key = bytes(v4).decode()
                        elif f4 == 2:
                            val = parse_attr(v4)
                    if key not in ("_class", "_output_shapes"):
                        nd["attr"][key] = val
            nodes.append(nd)
    return nodes


def main():
    out_dir = os.path.join(HERE, "v1_graph")
    os.makedirs(out_dir, exist_ok=True)
    for path in sorted(glob.glob(os.path.join(REF_MODELS, "*.meta"))):
        nodes = [n for n in parse_meta(path) if not n["name"].startswith("save/")]
        name = os.path.basename(path)[:-5]
        doc = dict(source="test_v1/model/%s.meta" % name, n_nodes_total=len(parse_meta(path)), nodes=nodes)
        raw = json.dumps(doc, separators=(",", ":"), sort_keys=True).encode()
        # mtime=0: byte-identical archives on every regeneration
        with open(os.path.join(out_dir, name + ".json.gz"), "wb") as fh:
            with gzip.GzipFile(fileobj=fh, mode="wb", mtime=0) as gz:
                gz.write(raw)
        print("%s: %d nodes kept (%d total), %d bytes json" % (name, len(nodes), doc["n_nodes_total"], len(raw)))


if __name__ == "__main__":
    main()
