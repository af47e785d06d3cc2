"""Inputs built from tests/golden/reference_kats.json that several suites share (tests/, tools/soak.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")


def groth16_regression_inputs(kats=None):
    """The shrunk inputs of /root/reference/tests/groth16_gadgets.proptest-regressions:7-15 as 32-byte little-endian
    records: -> dict(points (4, 32) Element encodings, scalars (4, 32): the three Fr values and the scalar byte array,
    fq (2, 32) field elements).  Fr / Fq are printed big-endian by the reference's Debug impls."""
    if kats is None:
        with open(GOLDEN) as f:
            kats = json.load(f)
    g = kats["groth16_gadget_regressions"]
    be = lambda h: np.frombuffer(bytes.fromhex(h)[::-1], dtype=np.uint8)
    points = np.array([list(bytes.fromhex(h)) for h in g["element_hex"].values()], dtype=np.uint8)
    scalars = np.stack([be(h) for h in g["fr_be_hex"].values()] +
                       [np.array(a, dtype=np.uint8) for a in g["scalar_arr_le_bytes"].values()])
    fq = np.stack([be(h) for h in g["fq_be_hex"].values()])
    return {"points": points, "scalars": scalars, "fq": fq}
