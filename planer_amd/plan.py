"""Plan compiler: peephole fusion of a planer flow (pure host logic, no GPU).

The reference interprets the flow one layer at a time (net.py:37-72).  On
MI355X the HBM-bound layers that follow a convolution -- folded BatchNorm
(layer.py:125-127), residual Add (layer.py:93-95), ReLU / LeakyReLU
(layer.py:44-51) -- cost more than the bytes they compute on, so the plan
folds each chain  conv -> batchnorm -> [add] -> [relu|leakyrelu]  into the
conv kernel's epilogue (`conv_fused`).  A link is absorbed only when the
intermediate tensor has exactly one reader and one writer, so nothing a user
could observe disappears; the fused step sits where the LAST link of its
chain was, so a residual operand produced after the conv is still available.
"""
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple)) else [v]


def expand_steps(flow):
    """Chained steps [x, [l1, l2], y] -> one step per layer (net.py:46-50)."""
    steps = []
    for src, names, dst in flow:
        for pos, name in enumerate(_as_list(names)):
            steps.append((_as_list(src if pos == 0 else dst), name, dst))
    return steps


def fuse_flow(layers, flow, init_names, shapes):
    """-> (layers', flow', number_of_absorbed_steps).

    `shapes` maps tensor keys to shapes (from one eager pass); a residual add
    is folded only when both operands have the same known shape.
    """
    kinds = {name: (kind, para) for name, kind, para in layers}
    steps = expand_steps(flow)
    readers, writers = {}, {}
    for i, (srcs, _, dst) in enumerate(steps):
        for k in set(srcs):
            readers.setdefault(k, []).append(i)
        for k in _as_list(dst):
            writers[k] = writers.get(k, 0) + 1
    inits = set(init_names)
    consumed, fused_at, nfused = set(), {}, 0
    for i, (srcs, name, dst) in enumerate(steps):
        kind, para = kinds[name]
        if kind != "conv" or i in consumed or not isinstance(dst, str):
            continue
        chain, cur, stage = [i], dst, 0
        extra = {"scale": "None", "shift": "None", "res": "None", "act": ACT_NONE, "alpha": 0.0}
        while stage < 3:
            r = readers.get(cur, [])
            if len(r) != 1 or writers.get(cur, 0) != 1:
                break
            j = r[0]
            jsrcs, jname, jdst = steps[j]
            jkind, jpara = kinds[jname]
            if j in consumed or j <= chain[-1] or not isinstance(jdst, str):
                break
            # an in-place relu on one of the conv's inputs between here and j would
            # change what the delayed conv reads (layer.py:46 mutates its input)
            if any(kinds[steps[s][1]][0] == "relu" and steps[s][0][0] in srcs
                   for s in range(chain[-1] + 1, j) if s not in chain):
                break
            if (jkind == "batchnorm" and stage < 1 and len(jsrcs) == 3 and jsrcs[0] == cur
                    and jsrcs[1] in inits and jsrcs[2] in inits):
                extra["scale"], extra["shift"], stage = jsrcs[1], jsrcs[2], 1
            elif (jkind == "add" and stage < 2 and len(jsrcs) == 2 and jsrcs.count(cur) == 1
                  and shapes.get(jsrcs[0]) is not None
                  and tuple(shapes.get(jsrcs[0])) == tuple(shapes.get(jsrcs[1]) or ())):
                extra["res"], stage = jsrcs[1 - jsrcs.index(cur)], 2
            elif jkind == "relu":
                extra["act"], stage = ACT_RELU, 3
            elif jkind == "leakyrelu":
                extra["act"], extra["alpha"], stage = ACT_LEAKY, jpara.get("alpha", 0.2), 3
            else:
                break
            chain.append(j)
            cur = jdst
        if len(chain) > 1:
            consumed.update(chain)
            fused_at[chain[-1]] = (srcs, name, para, extra, cur)
            nfused += len(chain) - 1
    body, out_flow, seen = [], [], set()

    def add_layer(entry):
        if entry[0] not in seen:
            seen.add(entry[0])
            body.append(entry)

    for i, (srcs, name, dst) in enumerate(steps):
        if i in fused_at:
            csrcs, cname, cpara, extra, out = fused_at[i]
            para = dict(cpara, act=extra["act"], alpha=extra["alpha"])
            args = [csrcs[0], csrcs[1], csrcs[2] if len(csrcs) > 2 else "None",
                    extra["scale"], extra["shift"], extra["res"]]
            add_layer([cname + "+", "conv_fused", para])
            out_flow.append([args, [cname + "+"], out])
        elif i not in consumed:
            kind, para = kinds[name]
            add_layer([name, kind, para])
            out_flow.append([srcs, [name], dst])
    return body, out_flow, nfused
