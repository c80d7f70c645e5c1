"""Shared assertions for the parity tests."""
import os

import numpy as np

import cases as K

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STRUCTURE_KEYS = ("tree_indices", "depths", "feature_indices", "feature_values", "is_numerics",
                  "inequality_directions", "categorical_values")
VALUE_KEYS = ("values", "edge_weights")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = K.BY_NAME[name]
    X, Xc, G, y = K.make_inputs(case)
    assert K.inputs_digest(X, Xc, G, y) == str(g["inputs_sha256"]), "input synthesis drifted from the fixture"
    return case, g, (X, Xc, G, y)


def assert_structure_equal(e, g, what=""):
    """Tree structure must be BIT-identical (split feature / threshold bits / directions / layout)."""
    for k in STRUCTURE_KEYS:
        a, b = np.asarray(e[k]), np.asarray(g[k])
        assert a.shape == b.shape, f"{what}{k}: shape {a.shape} != {b.shape}"
        if a.dtype.kind == "f":
            # Signed zeros are canonicalised on both sides: -0.0 and +0.0 are ONE order statistic (they compare equal), the
            # reference stores whichever of the two its std::sort happens to leave at the rank, the product stores +0.0; no
            # comparison `x > t` the model ever makes can tell them apart.  Every other threshold must match bit for bit.
            a = (a.astype(np.float32) + np.float32(0.0)).astype(np.float32)
            b = (b.astype(np.float32) + np.float32(0.0)).astype(np.float32)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
                f"{what}{k}: threshold bits differ at {np.argwhere(a.view(np.uint32) != b.view(np.uint32))[:5].tolist()}"
        else:
            assert np.array_equal(a, b), f"{what}{k}: differs at {np.argwhere(a != b)[:5].tolist()}"


def rel_err(a, b, scale):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), scale))) if a.size else 0.0


def assert_values_close(e, g, grad_scale, tol=1e-5, what=""):
    """Leaf values within `tol` relative: |a-b| <= tol * max(|b|, grad_scale) (north_star: 1e-5; SURVEY.md
    hard part 5 for why the floor is the gradient scale)."""
    for k in VALUE_KEYS:
        a, b = np.asarray(e[k]), np.asarray(g[k])
        assert a.shape == b.shape, f"{what}{k}: shape {a.shape} != {b.shape}"
        err = rel_err(a, b, grad_scale if k == "values" else 1.0)
        assert err <= tol, f"{what}{k}: rel err {err:.3g} > {tol}"
