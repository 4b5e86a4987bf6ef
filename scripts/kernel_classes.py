"""Which class (igemm_fwd / igemm_dgrad / igemm_wgrad) a conv launch of a rocprofv3 trace belongs to.

The class is decided by LAUNCH ORDER, the one thing every arithmetic (fp32 / bf16 / bf16x6) shares: a step launches every
forward conv (conv1_1 .. fc8), then walks backward with the filter gradient of a layer in front of its data gradient -- so the
conv launches of a step up to its first filter-gradient launch are forward, every later non-wgrad launch is a data gradient.
Where a kernel's name also carries the direction as a template argument (the fp32 kernels: BT = true -> data gradient) the two
must agree; `bt_from_name` reads that argument BY POSITION (round 3 looked at the LAST template argument, which since round 2 is
the stream-K flag of igemm2_kernel: every stream-K forward launch was booked as a data gradient).
"""
import re

# position of the direction flag BT in each fp32 kernel's template argument list (csrc/conv_igemm.hip, igemm_v2.h)
_BT_POS = {"igemm2_kernel": 6, "igemm_kernel": 4, "igemm_ws_kernel": 0, "conv_d64_kernel": 0}
_CONV = re.compile(r"\b((?:igemm|wgrad|conv_d64|conv_d4|conv_k32)\w*_kernel)\b(?:<([^>]*)>)?")   # (conv_d64_kernel<DG>, conv_d4_kernel, conv_k32_kernel: the direct conv1_2 / conv1_1 / fc8-dgrad kernels, round 5)
_SECOND_STAGE = ("streamk_fixup", "splitk_epilogue", "slab_sum", "rowmap")   # (streamk_fixup: profiles of rounds 2-3 and the opt-in bf16t kernels)


def conv_kernel(name):
    """(symbol, [template arguments]) of a conv launch, None for anything else (second stages included)."""
    if any(s in name for s in _SECOND_STAGE):
        return None
    m = _CONV.search(name)
    if not m:
        return None
    return m.group(1), [t.strip() for t in (m.group(2) or "").split(",") if t.strip()]


def bt_from_name(name):
    """True (data gradient) / False (forward) when the name says so, None when it does not (bf16 / bf16x6 / wgrad kernels)."""
    k = conv_kernel(name)
    if k is None or k[0] not in _BT_POS:
        return None
    sym, targs = k
    pos = _BT_POS[sym]
    if pos >= len(targs) or targs[pos] not in ("true", "false"):
        raise ValueError("cannot read BT of %s" % name)
    return targs[pos] == "true"


def classes_by_order(names):
    """Class of each launch of ONE step's conv launches, given in launch order."""
    out, backward = [], False
    for n in names:
        k = conv_kernel(n)
        if k is None:
            raise ValueError("not a conv launch: %s" % n)
        if k[0].startswith("wgrad"):
            backward = True
            out.append("igemm_wgrad")
            continue
        cls = "igemm_dgrad" if backward else "igemm_fwd"
        bt = bt_from_name(n)
        if bt is not None and bt != (cls == "igemm_dgrad"):
            raise ValueError("launch order says %s but the kernel's BT argument says otherwise: %s" % (cls, n))
        out.append(cls)
    return out
