#!/usr/bin/env python
"""Decode the constants baked into the reference's bundled SavedModel graph
(/root/reference/nmrgnn/models/baseline/saved_model.pb) with a minimal protobuf wire-format
reader (TensorFlow is not installed) and write them to tests/golden/savedmodel_constants.json.

These pin the oracle's RBF grid, peak standardisation vectors, noise sigma and dropout scale against
the reference's own artefact (the weight VALUES are absent from the bundle).
Run in the build container only; the JSON travels, this script's input does not.
"""
import json
import os
import struct
import sys

PB = "/root/reference/nmrgnn/models/baseline/saved_model.pb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "savedmodel_constants.json")


def varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def fields(b):
    """yield (field_number, wire_type, value) for one message"""
    i, n = 0, len(b)
    while i < n:
        key, i = varint(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield fn, wt, v


def parse_tensor(b):
    dtype, shape, content, fvals = None, [], b"", []
    for fn, wt, v in fields(b):
        if fn == 1:
            dtype = v
        elif fn == 2:
            for f2, w2, v2 in fields(v):
                if f2 == 2:
                    for f3, w3, v3 in fields(v2):
                        if f3 == 1:
                            shape.append(v3)
        elif fn == 4:
            content = v
        elif fn == 5:
            if wt == 2:
                fvals += list(struct.unpack(f"<{len(v)//4}f", v))
            else:
                fvals.append(struct.unpack("<f", v)[0])
    if dtype != 1:
        return None
    if content:
        vals = list(struct.unpack(f"<{len(content)//4}f", content))
    else:
        vals = fvals
    n = 1
    for s in shape:
        n *= s
    if len(vals) == 1 and n > 1:
        vals = vals * n
    return shape, vals


def parse_node(b):
    name, op, tensor = None, None, None
    for fn, wt, v in fields(b):
        if fn == 1:
            name = v.decode()
        elif fn == 2:
            op = v.decode()
        elif fn == 5:  # attr map entry
            key, val = None, None
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    key = v2.decode()
                elif f2 == 2:
                    val = v2
            if key == "value" and val is not None:
                for f3, w3, v3 in fields(val):
                    if f3 == 8:
                        tensor = parse_tensor(v3)
    return name, op, tensor


def walk_graph(graph_def, prefix, out):
    for fn, wt, v in fields(graph_def):
        if fn == 1:  # node
            name, op, tensor = parse_node(v)
            if op == "Const" and tensor is not None:
                out[prefix + name] = tensor
        elif fn == 2:  # library
            for f2, w2, v2 in fields(v):
                if f2 == 1:  # FunctionDef
                    fname = "?"
                    nodes = []
                    for f3, w3, v3 in fields(v2):
                        if f3 == 1:
                            for f4, w4, v4 in fields(v3):
                                if f4 == 1:
                                    fname = v4.decode()
                        elif f3 == 3:
                            nodes.append(v3)
                    for nb in nodes:
                        name, op, tensor = parse_node(nb)
                        if op == "Const" and tensor is not None:
                            out[f"{fname}/{name}"] = tensor


def main():
    b = open(PB, "rb").read()
    consts = {}
    for fn, wt, v in fields(b):
        if fn == 2:  # MetaGraphDef
            for f2, w2, v2 in fields(v):
                if f2 == 2:
                    walk_graph(v2, "", consts)
    if "--list" in sys.argv:
        for k, (shape, vals) in sorted(consts.items()):
            print(k, shape, vals[:4])
        return
    fn_inf = "__inference__wrapped_model_4657940"
    centers = next(v for k, (sh, v) in consts.items() if sh == [128] and "/" not in k)
    top_scalars = {k: v[0] for k, (sh, v) in consts.items() if sh == [] and "/" not in k}
    gap = [v for v in top_scalars.values() if 1e-3 < v < 2e-3]
    std = consts[fn_inf + "/gnn-model/mul_3/y"][1]
    avg = consts[fn_inf + "/gnn-model/mul_4/y"][1]
    tr = "__inference_gnn-model_layer_call_and_return_conditional_losses_4659631"
    out = {
        "source": "nmrgnn/models/baseline/saved_model.pb (reference bundle; graph only, no weight values)",
        "rbf_centers": centers,
        "rbf_gap_candidates": gap,
        "peak_std": std,
        "peak_avg": avg,
        "mask_threshold": consts[fn_inf + "/gnn-model/Greater/y"][1][0],
        "rbf_pow": consts[fn_inf + "/gnn-model/rbf-layer/pow/y"][1][0],
        "noise_stddev": consts[tr + "/gaussian_noise/random_normal/stddev"][1][0],
        "dropout_keep_scale": consts[tr + "/dropout/dropout/Const"][1][0],
        "dropout_rate": consts[tr + "/dropout/dropout/GreaterEqual/y"][1][0],
        "top_level_scalars": top_scalars,
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
