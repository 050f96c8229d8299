"""Source-level invariants of the product's host / device headers that no runtime test on a CPU box can see (no GPU here).

Speculative linearisation keeps a SECOND set of the linearisation's outputs (BatchDev members named `<first>2`, csrc/gfbe_device.h);
the kernels between two linearisations see the current set through lin_view (csrc/gfbe_devutil.h), which must swap every one of them,
and gfbe_host.cpp must point every one of them into the slab. A member added to one place and not the others would read a stale or
null array only in windows whose step was accepted — these checks fail first.
"""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ground-fusion2_amd", "csrc")


def _read(name):
    with open(os.path.join(CSRC, name)) as f:
        return f.read()


def _second_set_members():
    dev = _read("gfbe_device.h")
    body = dev[dev.index("struct BatchDev"):]
    body = body[:body.index("\n};")]
    names = re.findall(r"\*\s*([A-Za-z_][A-Za-z_0-9]*2)\b", body)
    members = set(re.findall(r"\*\s*([A-Za-z_][A-Za-z_0-9]*)\b", body))
    return sorted(n for n in set(names) if n[:-1] in members)


def test_second_set_is_declared():
    second = _second_set_members()
    assert len(second) >= 15 and "lm_hP2" in second and "gnss_cost2" in second, second


def test_lin_view_swaps_every_second_set_member():
    util = _read("gfbe_devutil.h")
    view = util[util.index("BatchDev lin_view("):]
    view = view[:view.index("return v;")]
    for name in _second_set_members():
        assert re.search(r"v\.%s\s*=\s*d\.%s\s*;" % (name[:-1], name), view), "lin_view does not swap " + name


def test_host_points_every_second_set_member_into_the_slab():
    host = _read("gfbe_host.cpp")
    for name in _second_set_members():
        assert re.search(r"\bd\.%s\s*=" % name, host), "gfbe_host.cpp never sets BatchDev::" + name


def test_options_default_and_binding_agree_on_the_speculative_pass():
    host = _read("gfbe_host.cpp")
    assert re.search(r"speculative_linearization\s*=\s*1\s*;", host)
    from importlib import import_module
    abi = import_module("ground-fusion2_amd.abi")
    assert abi.default_options().speculative_linearization == 1
