"""Binding B is a committed C file, and where the reference tree is present it is
compiled (-fsyntax-only) against the reference's OWN cariboulite_radio.h: the three replaced signatures
(cariboulite_radio.h:592-619), the struct members the stub touches and the sample-type layouts are machine-checked.
/root/reference does not exist on the GPU box: skipped there."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/software/libcariboulite/src"
STUB = os.path.join(ROOT, "tests", "binding_b", "cariboulite_radio_hip.c")


def test_stub_is_committed_and_matches_integration_md():
    src = open(STUB).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "tests/binding_b/cariboulite_radio_hip.c" in doc
    for fn in ("cariboulite_radio_read_samples", "cariboulite_radio_write_samples", "cariboulite_radio_get_native_mtu_size_samples"):
        assert fn in src and fn in doc


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_stub_compiles_against_the_reference_header():
    cmd = ["gcc", "-fsyntax-only", "-std=gnu11", "-Wall", "-Werror", "-I", REF, "-I", os.path.join(REF, "caribou_smi"),
           "-I", os.path.join(ROOT, "include"), STUB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # a wrong signature must be caught: flip one and expect a conflict with the reference's declaration
    bad = open(STUB).read().replace("size_t cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)",
                                    "int cariboulite_radio_get_native_mtu_size_samples(cariboulite_radio_state_st *radio)")
    r = subprocess.run(cmd[:-1] + ["-x", "c", "-"], input=bad, capture_output=True, text=True)
    assert r.returncode != 0 and "conflicting types" in r.stderr
