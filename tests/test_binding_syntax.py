"""towr_amd/csrc/towr_binding.h is the towr-typed half of the binding (ToTwrModel / ToTwrSchedule / ToTwrParams / ToTwrTerrain /
MakeDeviceConstraints(const NlpFormulation&)).  Eigen and ifopt are absent from this image, so it cannot be built and run here
(oracle/ref_dump --binding does that on a box that has them) -- but it CAN be type-checked: `g++ -fsyntax-only` of the header
against the REAL towr headers where they lie (/root/reference/towr/include, read-only), with type-level stand-ins for the
Eigen / ifopt names those headers mention (tests/towr_syntax_stub + tests/ifopt_stub: test infrastructure, nothing of the
reference is compiled or executed).  That catches what a header nobody compiles is most likely to have: a wrong member,
enum or method name of the reference.  Skipped where the reference is not present (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/towr/include"


def _check(header, tmp_path):
    src = tmp_path / "bind_check.cc"
    src.write_text('#include "%s"\n#ifndef TOWR_AMD_HAVE_RAPIDCSV\n#error "the CsvTerrain half was not seen"\n#endif\n'
                   'int main() { return 0; }\n' % header)
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "towr_syntax_stub"),
                           "-I" + os.path.join(ROOT, "tests", "ifopt_stub"), "-I" + REF, "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "towr_amd", "csrc"), str(src)], capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference headers are not on this box")
def test_binding_header_type_checks_against_the_reference_headers(tmp_path):
    r = _check(os.path.join(ROOT, "towr_amd", "csrc", "towr_binding.h"), tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    # the net has to hold: a misspelt reference method must be an error (i.e. the gated code really was compiled)
    text = open(os.path.join(ROOT, "towr_amd", "csrc", "towr_binding.h")).read()
    for good, bad in (("GetNominalStanceInBase", "GetNominalStanceInBaze"), ("force_limit_in_normal_direction_", "force_limit_normal_"),
                      ("towr::Parameters::EndeffectorRom", "towr::Parameters::EndEffectorRom")):
        assert good in text
        typo = tmp_path / "towr_binding_typo.h"
        typo.write_text(text.replace(good, bad))
        # (the header includes its neighbours by name: let the copy find them)
        for n in ("ifopt_adapter.h",):
            (tmp_path / n).write_text(open(os.path.join(ROOT, "towr_amd", "csrc", n)).read())
        r = _check(str(typo), tmp_path)
        assert r.returncode != 0 and bad in r.stderr, (good, r.stderr[-500:])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference headers are not on this box")
def test_ref_dump_driver_type_checks_too():
    """oracle/ref_dump/ref_dump.cc (the driver of the real reference, incl. its --binding comparison) cannot be built here
    either; the same type-level check keeps it honest against the reference's headers until a box with Eigen3 + ifopt
    compiles it for real."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DTWR_WITH_BINDING=1", "-I" + os.path.join(ROOT, "tests", "towr_syntax_stub"),
                        "-I" + os.path.join(ROOT, "tests", "ifopt_stub"), "-I" + REF, "-I" + os.path.join(REF, "towr", "terrain"),
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "towr_amd", "csrc"),
                        os.path.join(ROOT, "oracle", "ref_dump", "ref_dump.cc")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
