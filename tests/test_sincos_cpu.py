"""The device sin / cos (csrc/env_models.h: sincos_g) restated on the CPU with fma() - tests/aux/sincos_check.c - against
long double libm: absolute error below 1e-15 from |x| ~ 1 to 1e15, NaN for inf / NaN.  Runs without a GPU."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_half_angle_sincos_restatement_is_accurate_to_1e15(tmp_path):
    exe = str(tmp_path / "sincos_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(HERE, "aux", "sincos_check.c"), "-o", exe, "-lm"])
    out = subprocess.check_output([exe, "300000"], text=True)
    rows = re.findall(r"scale (\S+): old (\S+) (\S+)  new (\S+) (\S+)", out)
    assert len(rows) == 8, out
    for scale, _, _, es, ec in rows:
        assert float(es) < 1e-15 and float(ec) < 1e-15, (scale, es, ec)
    assert re.search(r"inf: -?nan -?nan", out) and re.search(r"nan: -?nan -?nan", out), out
