"""Measurement bookkeeping (CPU): which class a profiled conv launch is booked under, and that the committed per-class traffic
summaries (profiles/r04*_traffic.json, what bench.py's roofline.traffic is read from) add up to the per-kernel PMC tables they
were made from.  Round 3's summary looked at the LAST template argument of a kernel name -- the stream-K flag -- and booked every
stream-K forward launch as a data gradient (roofline.traffic wrong by 4.5x)."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from kernel_classes import bt_from_name, classes_by_order, conv_kernel  # noqa: E402

NS = "void (anonymous namespace)::"
ARGS = "((anonymous namespace)::IgemmArgs)"


def test_direction_is_read_by_position_not_from_the_last_argument():
    assert bt_from_name(NS + "igemm2_kernel<4, 2, 16, 4, 4, 3, false, 16, true>" + ARGS) is False   # stream-K FORWARD
    assert bt_from_name(NS + "igemm2_kernel<4, 2, 16, 4, 4, 3, true, 9, false>" + ARGS) is True     # plain-tile data gradient
    assert bt_from_name(NS + "igemm_kernel<256, 64, 16, 64, true>" + ARGS) is True
    assert bt_from_name(NS + "igemm_ws_kernel<false>" + ARGS) is False
    assert bt_from_name(NS + "igemm_bf16_kernel<256, 256, 128, true, true, 0>" + ARGS) is None     # storage flags, no direction
    assert bt_from_name(NS + "igemm_x6p_kernel<true, 0>" + ARGS) is None
    assert bt_from_name(NS + "wgrad2_kernel<true>((anonymous namespace)::WgradArgs)") is None
    assert conv_kernel("(anonymous namespace)::streamk_fixup_kernel((anonymous namespace)::IgemmArgs, int)") is None
    assert conv_kernel(NS + "maxpool_fwd_kernel<float, 4>(float const*)") is None


def test_classes_follow_launch_order_for_every_arithmetic():
    fwd = [NS + "igemm_kernel<256, 64, 16, 64, false>" + ARGS, NS + "igemm2_kernel<4, 2, 16, 4, 4, 3, false, 9, true>" + ARGS,
           NS + "igemm2_kernel<4, 2, 16, 4, 4, 3, false, 1, false>" + ARGS, NS + "igemm_ws_kernel<false>" + ARGS]
    bwd = [NS + "wgrad_ws_kernel<64>(W)", NS + "igemm_kernel<256, 64, 16, 64, true>" + ARGS, NS + "wgrad2_kernel<true>(W)",
           NS + "igemm2_kernel<4, 2, 16, 4, 4, 3, true, 1, true>" + ARGS, NS + "igemm_ws_kernel<true>" + ARGS, NS + "wgrad2_kernel<false>(W)"]
    assert classes_by_order(fwd + bwd) == ["igemm_fwd"] * 4 + ["igemm_wgrad", "igemm_dgrad", "igemm_wgrad", "igemm_dgrad", "igemm_dgrad", "igemm_wgrad"]
    bf = [NS + "igemm_bf16_kernel<256, 256, 128, true, true, 0>" + ARGS] * 3 + [NS + "wgrad_bf16_kernel<256, 256, 32, true, true>(W)",
                                                                               NS + "igemm_bf16_kernel<256, 256, 128, true, true, 0>" + ARGS]
    assert classes_by_order(bf) == ["igemm_fwd"] * 3 + ["igemm_wgrad", "igemm_dgrad"]
    with pytest.raises(ValueError):   # a forward-named fp32 kernel behind a filter gradient: the two sources disagree
        classes_by_order([NS + "wgrad2_kernel<true>(W)", NS + "igemm_ws_kernel<false>" + ARGS])
    with pytest.raises(ValueError):
        classes_by_order([NS + "maxpool_fwd_kernel<float, 4>(x)"])


TRAFFIC = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) if int(re.search(r"r(\d+)", os.path.basename(p)).group(1)) >= 4)


@pytest.mark.parametrize("path", TRAFFIC, ids=[os.path.basename(p) for p in TRAFFIC])
def test_per_class_traffic_equals_the_per_kernel_table(path):
    t = json.load(open(path))
    for key in ("read_bytes", "write_bytes"):
        by_class = sum(c[key] for c in t["per_step"].values())
        by_kernel = sum(k[key] for k in t["per_kernel"].values())
        assert by_class == pytest.approx(by_kernel, rel=1e-9)
        for k in t["per_kernel"].values():
            assert sum(c[key] for c in k["by_class"].values()) == pytest.approx(k[key], rel=1e-9)
    # ... and the text table committed beside it (same PMC passes) says the same, kernel by kernel
    table = path.replace("_traffic.json", "_pmc_conv_kernels.txt")
    rows, name = {}, None
    for line in open(table):
        m = re.match(r"(\S.*?)\s+launches/step", line)
        if m:
            name = m.group(1).strip()
        m = re.search(r"HBM traffic/step: read ([\d.]+) GB .* write ([\d.]+) GB", line)
        if m and name:
            rows[name] = (float(m.group(1)), float(m.group(2)))
    assert rows, table
    for name, (rd, wr) in rows.items():
        k = t["per_kernel"][name]
        assert k["read_bytes"] / 1e9 == pytest.approx(rd, abs=0.006) and k["write_bytes"] / 1e9 == pytest.approx(wr, abs=0.006)
    # forward launches of a stream-K plan must be booked as forward: the class holds as many launches as the network has layers
    assert t["per_step"]["igemm_fwd"]["launches"] >= 16 and t["per_step"]["igemm_wgrad"]["launches"] == 16


def test_a_traffic_summary_of_this_round_is_committed():
    assert TRAFFIC, "profiles/r04*_traffic.json missing: run scripts/make_profiles.sh + scripts/summarize_profiles.py"
