"""Plan compiler: peephole fusion of a planer flow (pure host logic, no GPU).

The reference interprets the flow one layer at a time (net.py:37-72).  On
MI355X the HBM-bound layers that follow a convolution -- folded BatchNorm
(layer.py:125-127), residual Add (layer.py:93-95), ReLU / LeakyReLU
(layer.py:44-51) -- cost more than the bytes they compute on, so the plan
folds each chain  conv -> batchnorm -> [add] -> [relu|leakyrelu] -> [add]  into
the conv kernel's epilogue (`conv_fused`; the residual goes before the
activation in ResNet's blocks and after it in YOLO-v3's, never both).  A link is absorbed only when the
intermediate tensor has exactly one reader and one writer, so nothing a user
could observe disappears; the fused step sits where the LAST link of its
chain was, so a residual operand produced after the conv is still available.
"""
import os

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
ACT_RES_AFTER = 16      # OR-ed into `act`: the residual is added after the activation


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
        while stage < 4:
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
            elif (jkind == "add" and (stage < 2 or (stage == 3 and extra["res"] == "None"))
                  and len(jsrcs) == 2 and jsrcs.count(cur) == 1
                  and shapes.get(jsrcs[0]) is not None
                  and tuple(shapes.get(jsrcs[0])) == tuple(shapes.get(jsrcs[1]) or ())):
                extra["res"] = jsrcs[1 - jsrcs.index(cur)]
                if stage == 3:
                    extra["act"] |= ACT_RES_AFTER
                stage = 2 if stage < 2 else 4
            elif jkind == "relu" and stage < 3:
                extra["act"], stage = ACT_RELU, 3
            elif jkind == "leakyrelu" and stage < 3:
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


# ---- activation layout assignment -------------------------------------------------------------
# Kinds that have a channel-quad (Q4) kernel besides conv (planer_amd/q4.py).  A step of one of
# these kinds runs in Q4 when one of its activation inputs already is Q4 -- layouts are "sticky"
# downstream of a conv -- and everything else reads NCHW, with a conversion step inserted where a
# value is needed in the layout it was not produced in (converted copies are cached per value).
Q4_POINTWISE = ("maxpool", "averagepool", "gap", "upsample", "batchnorm", "relu", "leakyrelu", "sigmoid",
                "add", "concat")
TO_Q4, FROM_Q4 = "@to_q4", "@from_q4"


def _is4d(shapes, key):
    s = shapes.get(key)
    return s is not None and len(s) == 4


def q4_conv_ok(srcs, para, inits, shapes):
    """A conv step can take the Q4 kernel: constant 4-D filter (and constant bias / scale / shift),
    4-D input, symmetric pads, and groups that do not split a channel quad."""
    if len(srcs) < 2 or srcs[1] not in inits or not _is4d(shapes, srcs[1]) or not _is4d(shapes, srcs[0]):
        return False
    if any(k != "None" and k not in inits for k in srcs[2:5]):
        return False
    cout, cin_g = shapes[srcs[1]][0], shapes[srcs[1]][1]
    group = int(para.get("group", 1))
    pads = list(para.get("pads", (0, 0, 0, 0)))
    if len(pads) == 4 and (pads[0] != pads[2] or pads[1] != pads[3]):
        return False
    return group == 1 or (cin_g % 4 == 0 and (cout // group) % 4 == 0)


def _q4_pointwise_ok(kind, srcs, para, inits, shapes):
    acts = [k for k in srcs if k != "None" and k not in inits]
    if not acts or not all(_is4d(shapes, k) for k in acts):
        return False
    c = shapes[acts[0]][1]
    if kind == "add":
        return len(srcs) == 2 and len(acts) == 2 and tuple(shapes[srcs[0]]) == tuple(shapes[srcs[1]])
    if kind == "concat":
        return para.get("axis", 0) in (1, -3) and len(acts) == len(srcs) and all(shapes[k][1] % 4 == 0 for k in srcs)
    if kind == "sigmoid":
        return c % 4 == 0
    if kind == "batchnorm":
        return len(srcs) == 3 and srcs[1] in inits and srcs[2] in inits
    if kind == "upsample":
        return len(srcs) == 2 and srcs[1] in inits and para.get("mode", "nearest") == "nearest"
    return len(acts) == 1


# Rough device rates for the go / no-go estimate below: the Q4 conv kernel saves ~15 % of a conv's
# time at ~100 TFLOP/s; a layout conversion reads and writes its tensor once at ~4 TB/s.
_Q4_CONV_GAIN_S_PER_FLOP = 0.15 / 100e12
_CONVERT_S_PER_BYTE = 2.0 / 4e12


def _nbytes(shape):
    n = 4
    for d in shape:
        n *= d
    return n


def assign_layouts(body, flow, init_names, shapes, force=False):
    """-> (body', flow', number of Q4 steps).  Rewrites conv / conv_fused steps to `conv_q4` and the
    HBM-bound layers that follow them to their `*_q4` kinds, inserting `to_q4` / `from_q4` steps at
    the edges.  The program's observable values (its last step's outputs) stay NCHW.

    Unless `force`, the rewrite is dropped (-> the input program, 0) when the conversions it needs
    would cost more than the convs gain -- e.g. a lone conv whose large output has to be handed
    back as NCHW right away."""
    kinds = {name: (kind, para) for name, kind, para in body}
    inits = set(init_names)
    steps = expand_steps(flow)
    q4 = set()          # keys whose primary copy is Q4
    copies = {}         # (key, layout) -> key of the cached converted copy
    out_body, out_flow, seen = [], [], set()
    nq4 = 0

    def add_layer(entry):
        if entry[0] not in seen:
            seen.add(entry[0])
            out_body.append(list(entry))

    est = {"gain": 0.0, "cost": 0.0}

    def need(key, want_q4):
        if key == "None" or key in inits or (key in q4) == want_q4:
            return key
        ck = copies.get((key, want_q4))
        if ck is None:
            if shapes.get(key.split("@")[0]) is not None:
                est["cost"] += _nbytes(shapes[key.split("@")[0]]) * _CONVERT_S_PER_BYTE
            ck = key + ("@q4" if want_q4 else "@nchw")
            conv_name = TO_Q4 if want_q4 else FROM_Q4
            add_layer([conv_name, conv_name[1:], {}])
            out_flow.append([[key], [conv_name], ck])
            copies[(key, want_q4)] = ck
            if want_q4:
                q4.add(ck)
        return ck

    def drop_copies(key):
        for lay in (True, False):
            copies.pop((key, lay), None)

    last = len(steps) - 1
    for i, (srcs, name, dst) in enumerate(steps):
        kind, para = kinds[name]
        single = isinstance(dst, str)
        as_q4 = False
        if kind in ("conv", "conv_fused") and single and q4_conv_ok(srcs, para, inits, shapes):
            as_q4 = True
            full = list(srcs) + ["None"] * (6 - len(srcs))
            res = full[5]
            if res != "None" and (res in inits or not _is4d(shapes, res)):
                as_q4 = False
            else:
                # 1..3 input channels arriving as NCHW (the stem): the row-packed kernel reads a padded
                # NHWC copy it makes itself -- K without the 4th padding channel, no to_q4 step
                k = shapes[srcs[1]]
                rowpack = (os.environ.get("PLANER_HIP_ROWPACK", "1") != "0"
                           and full[0] not in q4 and k[1] < 4 and int(para.get("group", 1)) == 1
                           and list(para.get("dilations", (1, 1))) == [1, 1])
                args = [need(full[0], not rowpack)] + full[1:5] + [need(res, True)]
                new_kind = "conv_q4"
                if rowpack:
                    para = dict(para, rowpack=True)
                if shapes.get(dst) is not None:
                    k = shapes[srcs[1]]
                    est["gain"] += 2.0 * (_nbytes(shapes[dst]) / 4) * k[1] * k[2] * k[3] * _Q4_CONV_GAIN_S_PER_FLOP
        elif kind in Q4_POINTWISE and single and _q4_pointwise_ok(kind, srcs, para, inits, shapes) \
                and any(k in q4 for k in srcs):
            as_q4 = True
            args = [need(k, True) for k in srcs]
            new_kind = kind + "_q4"
        if not as_q4:
            args = [need(k, False) for k in srcs]
            new_kind = kind
        if kind == "relu":                       # in place (layer.py:46): cached copies of the input go stale
            drop_copies(srcs[0])
        out_key = dst
        produces_q4 = as_q4 and kind != "gap"
        if produces_q4 and i == last:
            out_key = dst + "@q4"                # the program's result is handed back as NCHW below
        add_layer([name, new_kind, para])
        out_flow.append([args, [name], out_key])
        for k in _as_list(dst):
            q4.discard(k)
            drop_copies(k)
        if produces_q4:
            q4.add(out_key)
            nq4 += 1
            if i == last:
                add_layer([FROM_Q4, FROM_Q4[1:], {}])
                out_flow.append([[out_key], [FROM_Q4], dst])
                if shapes.get(dst) is not None:
                    est["cost"] += _nbytes(shapes[dst]) * _CONVERT_S_PER_BYTE
        elif as_q4:
            nq4 += 1
    if not force and est["cost"] > est["gain"]:
        return [list(b) for b in body], [[list(srcs), [name], dst] for srcs, name, dst in steps], 0
    return out_body, out_flow, nq4


# ---- Winograd chaining ------------------------------------------------------------------------------
# A conv_q4 step that runs F(4x4,3x3) (w_layout 7) is three kernels: input transform (x -> V), the 36
# grouped GEMMs (V, U -> M) and output transform + fused tail (M -> y).  When the y of one such conv
# feeds another, "M -> y -> V'" can be ONE kernel that keeps y on chip (csrc/wino4_chain_kernel.h), and
# when nothing else reads y it is never written at all.  `chain_winograd` makes the stages explicit plan
# steps and merges the out / in pairs.  Kinds that only read their inputs (no in-place update): a
# Winograd input transform may be hoisted over them.
_PURE_READERS = ("conv_q4", "wino4_in", "wino4_gemm", "wino4_out", "wino4_chain", "wino43_in", "wino43_gemm", "wino43_out",
                 "wino43_chain", "conv1x1_wino_in", "conv_q4_pair", "add_q4", "maxpool_q4",
                 "averagepool_q4", "gap_q4", "upsample_q4", "concat_q4", "upconcat_q4", "batchnorm_q4",
                 "leakyrelu_q4", "sigmoid_q4", "from_q4")
WINO4_LAYOUT = 7
# w_layout -> stage-kind prefix: staged F(4x4,3x3), and the mixed-tile form for maps of 7 / 14 / 21 a side (q4.Wino43*)
STAGED_LAYOUTS = {7: "wino4", 11: "wino43"}


def chain_winograd(body, flow, supported=lambda key: True, chain=True):
    """-> (body', flow', number of chained pairs).  `flow` holds one layer per step (as made by
    assign_layouts / Net._prepare_filters); `supported(key)` says whether the LDS transform kernel
    can take the activation `key` (whole planes must fit a workgroup's LDS)."""
    kinds = {b[0]: b for b in body}
    steps = []                      # [srcs, name, kind, para, dst]
    for src, names, dst in flow:
        name = names[0] if isinstance(names, (list, tuple)) else names
        srcs = list(src) if isinstance(src, (list, tuple)) else [src]
        _, kind, para = kinds[name]
        if kind == "conv_q4" and para.get("w_layout") in STAGED_LAYOUTS and isinstance(dst, str):
            full = srcs + ["None"] * (6 - len(srcs))
            tail = {k: para[k] for k in ("act", "alpha") if k in para}
            pre = STAGED_LAYOUTS[para["w_layout"]]
            steps.append([[full[0]], name + "@in", pre + "_in", {}, name + "@V"])
            steps.append([[name + "@V", full[1]], name + "@gemm", pre + "_gemm", {}, name + "@M"])
            steps.append([[name + "@M"] + full[2:6], name + "@out", pre + "_out", tail, dst])
        else:
            steps.append([srcs, name, kind, para, dst])
    nchained = 0
    if chain:
        last_dsts = set(_as_list(steps[-1][4])) if steps else set()
        i = 0
        while i < len(steps):
            srcs, name, kind, para, dst = steps[i]
            if kind in ("wino4_out", "wino43_out") and (kind == "wino43_out" or supported(dst)):
                pre = kind[:-len("_out")]
                readers = [j for j in range(i + 1, len(steps)) if dst in steps[j][0]]
                # overwritten later under the same key?  then only readers before that point count
                rewrite = [j for j in range(i + 1, len(steps)) if dst in _as_list(steps[j][4])]
                stop = rewrite[0] if rewrite else len(steps)
                readers = [j for j in readers if j <= stop]
                j = next((j for j in readers if steps[j][2] == pre + "_in" and steps[j][0] == [dst]), None)
                if j is not None and all(steps[k][2] in _PURE_READERS for k in readers if k < j):
                    vkey = steps[j][4]
                    keep = len(readers) > 1 or dst in last_dsts
                    base = name[:-len("@out")]
                    steps[i] = [srcs, base + "@chain", pre + "_chain", dict(para, keep_y=keep), [dst, vkey] if keep else vkey]
                    del steps[j]
                    nchained += 1
            i += 1
    out_body, seen = [], set()
    for srcs, name, kind, para, dst in steps:
        if name not in seen:
            seen.add(name)
            out_body.append([name, kind, para])
    return out_body, [[srcs, [name], dst] for srcs, name, kind, para, dst in steps], nchained


# ---- 1x1 conv -> staged Winograd conv ----------------------------------------------------------------------
def fuse_conv1x1_wino_in(body, flow, kshape=lambda key: None, small=lambda key: True):
    """-> (body', flow', number of fused pairs).  A direct channel-quad 1x1 conv (w_layout 2, stride 1, no padding, group 1,
    no residual) whose ONLY reader is the input-transform stage of a staged F(4x4,3x3) conv (`wino4_in`, made explicit by
    chain_winograd) becomes one `conv1x1_wino_in` step that writes that stage's V (q4.Conv1x1WinoIn): the 1x1 -> 3x3 pairs of a
    Darknet block.  `small(key)` says whether the activation is small enough for the launch saved to matter (the fused kernel
    computes the 1x1 conv on overlapping patches: 1.9x its multiplies)."""
    kinds = {b[0]: b for b in body}
    steps = [[list(src) if isinstance(src, (list, tuple)) else [src], names[0] if isinstance(names, (list, tuple)) else names, dst]
             for src, names, dst in flow]
    readers, writers = {}, {}
    for i, (srcs, name, dst) in enumerate(steps):
        for k in set(srcs):
            readers.setdefault(k, []).append(i)
        for k in _as_list(dst):
            writers.setdefault(k, []).append(i)
    last_dsts = set(_as_list(steps[-1][2])) if steps else set()
    drop, repl, nfused = set(), {}, 0
    for i, (srcs, name, dst) in enumerate(steps):
        if kinds[name][1] != "wino4_in" or len(srcs) != 1:
            continue
        y = srcs[0]
        if len(writers.get(y, [])) != 1 or readers.get(y, []) != [i] or y in last_dsts:
            continue
        j = writers[y][0]
        csrcs, cname, cdst = steps[j]
        _, ckind, cpara = kinds[cname]
        full = csrcs + ["None"] * (6 - len(csrcs))
        if j >= i or j in drop or ckind != "conv_q4" or cpara.get("w_layout") != 2 or not isinstance(cdst, str) or full[5] != "None":
            continue
        k = kshape(full[1])
        if (k is None or tuple(k[2:]) != (1, 1) or k[0] % 4 or int(cpara.get("group", 1)) != 1
                or [int(v) for v in cpara.get("strides", (1, 1))] != [1, 1] or [int(v) for v in cpara.get("dilations", (1, 1))] != [1, 1]
                or any(int(v) for v in cpara.get("pads", (0, 0, 0, 0))) or int(cpara.get("act", 0)) & ~3 or not small(y)):
            continue
        fname = cname + "@v4"
        repl[j] = (full[:5], fname, "conv1x1_wino_in", {"act": int(cpara.get("act", 0)), "alpha": float(cpara.get("alpha", 0.0)), "wino": 4}, dst)
        drop.add(i)
        nfused += 1
    out = []
    for i, (srcs, name, dst) in enumerate(steps):
        if i in drop:
            continue
        if i in repl:
            out.append(repl[i])
        else:
            out.append((srcs, name, kinds[name][1], kinds[name][2], dst))
    out_body, seen = [], set()
    for srcs, name, kind, para, dst in out:
        if name not in seen:
            seen.add(name)
            out_body.append([name, kind, para])
    return out_body, [[srcs, [name], dst] for srcs, name, kind, para, dst in out], nfused


# ---- sibling convolutions --------------------------------------------------------------------------------
# Where a graph forks into two direct channel-quad convs on the same tensor -- a ResNet block that changes resolution:
# the stride-2 3x3 conv and the 1x1 stride-2 projection -- both run in ONE launch (q4.ConvQ4Pair,
# csrc/conv_q4_kernel.h conv_q4_pair_kernel).  The second conv moves up to the first one's place; it only needs the
# shared input and constants, so that is legal unless something rewrites the input in place in between.
_IN_PLACE = ("relu", "relu_q4", "clip", "erf", "instancenormalization")


def pair_sibling_convs(body, flow, kshape=lambda key: None):
    """-> (body', flow', number of pairs).  One layer per step, as made by Net._prepare_filters.  `kshape(key)` gives
    the OIHW shape of a filter key; only the projection pattern is paired -- equal strides > 1, one of the two a 1x1 --
    because both convs then have the same output map and neither wants a split-K plan of its own."""
    kinds = {b[0]: b for b in body}
    steps = [[list(src) if isinstance(src, (list, tuple)) else [src], names[0] if isinstance(names, (list, tuple)) else names, dst]
             for src, names, dst in flow]

    def direct(i):
        srcs, name, dst = steps[i]
        _, kind, para = kinds[name]
        full = srcs + ["None"] * (6 - len(srcs))
        return (kind == "conv_q4" and para.get("w_layout") == 2 and isinstance(dst, str) and full[5] == "None"
                and int(para.get("group", 1)) == 1 and [int(v) for v in para.get("dilations", (1, 1))] == [1, 1]
                and not para.get("rowpack") and not (int(para.get("act", 0)) & ~3))

    used, npairs, out = set(), 0, []
    for i in range(len(steps)):
        if i in used:
            continue
        srcs, name, dst = steps[i]
        j = None
        if direct(i):
            for k in range(i + 1, len(steps)):
                ks, kname, kdst = steps[k]
                if srcs[0] in ks and kinds[kname][1] in _IN_PLACE:
                    break                                   # the shared input is rewritten: later readers see other values
                if srcs[0] in _as_list(kdst):
                    break
                # the flow's LAST step defines the program's result (net.py:72): hoisting it would make another step last
                if k not in used and k != len(steps) - 1 and i != len(steps) - 1 and direct(k) and ks[0] == srcs[0]:
                    p1, p2 = kinds[name][2], kinds[kname][2]
                    k1, k2 = kshape(srcs[1]), kshape(ks[1])
                    st1, st2 = [int(v) for v in p1.get("strides", (1, 1))], [int(v) for v in p2.get("strides", (1, 1))]
                    if (k1 is not None and k2 is not None and st1 == st2 and min(st1) > 1
                            and min(k1[2] * k1[3], k2[2] * k2[3]) == 1):
                        j = k
                        break
        if j is None:
            out.append((srcs, name, kinds[name][1], kinds[name][2], dst))
            continue
        s2, n2, d2 = steps[j]
        f1, f2 = srcs + ["None"] * (6 - len(srcs)), s2 + ["None"] * (6 - len(s2))
        pname = name + "&" + n2
        out.append(([f1[0]] + f1[1:5] + f2[1:5], pname, "conv_q4_pair",
                    {"para1": dict(kinds[name][2]), "para2": dict(kinds[n2][2])}, [dst, d2]))
        used.add(j)
        npairs += 1
    out_body, seen = [], set()
    for srcs, name, kind, para, dst in out:
        if name not in seen:
            seen.add(name)
            out_body.append([name, kind, para])
    return out_body, [[srcs, [name], dst] for srcs, name, kind, para, dst in out], npairs
