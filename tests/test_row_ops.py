"""The lane-spread field arithmetic (decaf377_amd/csrc/row_ops.hpp): its Python model on the CPU (exactness against a * b
mod q, every accumulator bound at the contract's worst case), and on the GPU the kernel's products and group operations
against that model / against the formulas on Python integers (tools/row_proto.hip is the harness)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_model_exact_and_bounded():
    """tools/row_model.py: 3 000 random and worst-case products through the same steps as the kernel (16-lane row, fold
    constants, three carry passes) equal a * b mod q, and no 64-bit accumulator or 32-bit carry overflows for operands at
    the tight / lazy contract bounds."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "row_model.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ROW_MODEL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_wave_inversion_model_exact_and_bounded():
    """tools/inv_wave_model.py: the wave's inversion (row_ops.hpp fe_invert_wave) step for step on integers -- ~7 900 structured
    and random residues give pow(c, -1, q), with every product sum inside int64, every limb inside the range the next round
    assumes, d and e inside (-(round + 2) q, q] and g = 0 after 25 rounds."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "inv_wave_model.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "INV_WAVE_MODEL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_row_constants_are_current():
    """decaf377_amd/csrc/row_constants.inc is what tools/gen_row_constants.py generates (fold residues, subtraction biases)."""
    path = os.path.join(ROOT, "decaf377_amd", "csrc", "row_constants.inc")
    before = open(path).read()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_row_constants.py")], capture_output=True, text=True, timeout=120)
    after = open(path).read()
    if after != before:
        open(path, "w").write(before)
    assert r.returncode == 0 and after == before, "row_constants.inc is stale: run tools/gen_row_constants.py"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import row_model as rm
    k = int(before.split("ROW_SUB_TIGHT = ")[1].split(" q")[0])
    limbs = [int(x, 16) for x in before.split("ROW_SUB_TIGHT[16] = {")[1].split("}")[0].replace("u", "").split(",")]
    assert sum(v << (28 * j) for j, v in enumerate(limbs[:10])) == k * rm.Q and all(v >= (1 << 28) + (1 << 12) for v in limbs[:9])


@pytest.mark.gpu
def test_row_ops_on_the_gpu():
    """row_mul bit-identical to the model on 4 096 operand pairs (tight, lazy, worst case); rq_double_neg / rq_add against
    the group formulas on integers for 200 random inputs (and chained doublings); outputs tight, lanes 10..15 zero; the
    wave's inversion (fe_invert_wave) equal to the lane's on 2 048 random, small and edge values, x * (1/x) = 1, 0 -> 0."""
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    so = os.path.join(ROOT, "build", "row_proto.so")
    deps = [os.path.join(ROOT, "tools", "row_proto.hip")] + [os.path.join(ROOT, "decaf377_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "decaf377_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(d_) > os.path.getmtime(so) for d_ in deps):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                               deps[0], "-o", so], timeout=900)
    for tool, ok in (("row_proto.py", "0 differ from the model"), ("row_point_check.py", "ROW_POINT_OK"), ("row_invert_check.py", "ROW_INVERT_OK")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0 and ok in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
