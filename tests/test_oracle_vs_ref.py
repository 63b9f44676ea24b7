"""Where the reference binary is available (oracle/_ref/yak, built by `make -C oracle ref` from
/root/reference or carried prebuilt to the GPU box), run it next to the oracle CLI."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "yak")
YKO = os.path.join(ROOT, "oracle", "yko")
SYN = os.path.join(ROOT, "tools", "yaksynth")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="prebuilt reference binary not present")

COMBOS = [["-k31"], ["-k31", "-b20"], ["-k31", "-b25"], ["-k21", "-K50k", "-t3"], ["-k41"], ["-k63", "-b24"],
          ["-k27", "-p12", "-b26"], ["-k31", "-b22", "-H40"]]


@pytest.mark.parametrize("args", COMBOS, ids=lambda a: "".join(a))
def test_cli_bytes_identical(args, tmp_path, oracle):
    fq = str(tmp_path / "r.fq")
    subprocess.check_call([SYN, "-n", "4000", "-l", "150", "-g", "20000", "-s", "77", "-o", fq])
    a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
    subprocess.run([REF, "count"] + args + ["-o", a, fq], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([YKO, "count"] + args + ["-o", b, fq], check=True, stderr=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_two_input_files(tmp_path, oracle):
    f1, f2 = str(tmp_path / "1.fq"), str(tmp_path / "2.fq")
    subprocess.check_call([SYN, "-n", "4000", "-g", "20000", "-s", "5", "-o", f1])
    subprocess.check_call([SYN, "-n", "3000", "-g", "20000", "-s", "5", "-e", "0.01", "-o", f2])
    a, b = str(tmp_path / "a.yak"), str(tmp_path / "b.yak")
    subprocess.run([REF, "count", "-b24", "-o", a, f1, f2], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([YKO, "count", "-b24", "-o", b, f1, f2], check=True, stderr=subprocess.DEVNULL)
    assert open(a, "rb").read() == open(b, "rb").read()
