#!/usr/bin/env python3
"""Golden vectors for the pose-error metric, produced by the REFERENCE's own code.

The reference measures a pose estimate against the truth with tools/evaluate_rpe.py: ominus (relative transformation),
compute_angle (rotation angle of its 3x3 block) and compute_distance (norm of its translation) -- tools/evaluate_rpe.py:138-173;
src/exp1/exp1_2.cpp:167-171 prints the same two numbers.  That file is Python 2 as a whole (print statements in its command-line
part), so it cannot be imported here; this script reads it where it lies, under /root/reference, and executes ONLY those three
function definitions (plain numpy, valid Python 3) -- nothing of the reference is copied into this repository, only the
numbers its functions return:

    tests/golden/pose_error_golden.json   [{"A": 16 doubles, "B": 16 doubles, "angle": rad, "distance": m}, ...]

tests/test_oracle_math.py::test_pose_error_matches_the_reference_tool checks oracle/icp_oracle.c::orc_pose_error (the metric of
every parity test) against them.  Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_pose_error_golden.py
"""
import ast
import hashlib
import json
import os
import re

import numpy
from scipy.spatial.transform import Rotation

REF = "/root/reference/tools/evaluate_rpe.py"
HERE = os.path.dirname(os.path.abspath(__file__))


# The reference is untrusted input: before any of its text is executed, every extracted definition must parse to NOTHING but
# arithmetic on its arguments and calls of a fixed set of numpy functions (exec() of extracted text runs with the
# developer's privileges).  No imports, no attribute access outside numpy / numpy.linalg, no names beyond the whitelist, no
# builtins at run time.  The SHA-256 of what was executed is recorded in the JSON.
_ALLOWED_NODES = (ast.Module, ast.FunctionDef, ast.arguments, ast.arg, ast.Return, ast.Expr, ast.Constant, ast.Name, ast.Load, ast.Store,
                  ast.Call, ast.Attribute, ast.BinOp, ast.UnaryOp, ast.Add, ast.Sub, ast.Mult, ast.Div, ast.USub, ast.Subscript, ast.Slice,
                  ast.Tuple, ast.Assign, ast.keyword)
_ALLOWED_NUMPY = {"dot", "trace", "arccos", "min", "max", "linalg", "inv", "norm"}
_ALLOWED_NAMES = {"numpy", "min", "max"}


def _vet(name, text):
    tree = ast.parse(text)
    if len(tree.body) != 1 or not isinstance(tree.body[0], ast.FunctionDef) or tree.body[0].name != name or tree.body[0].decorator_list:
        raise SystemExit(f"{name}: expected exactly one plain function definition")
    args = {a.arg for a in tree.body[0].args.args}
    for node in ast.walk(tree):
        if not isinstance(node, _ALLOWED_NODES):
            raise SystemExit(f"{name}: {type(node).__name__} is not allowed in a reference function that gets executed")
        if isinstance(node, ast.Attribute) and node.attr not in _ALLOWED_NUMPY:
            raise SystemExit(f"{name}: attribute .{node.attr} is not on the whitelist")
        if isinstance(node, ast.Name) and node.id not in _ALLOWED_NAMES | args:
            raise SystemExit(f"{name}: name {node.id} is not on the whitelist")
    return tree


def reference_functions():
    src = open(REF).read()
    ns = {"numpy": numpy, "__builtins__": {"min": min, "max": max}}
    digest = hashlib.sha256()
    for name in ("ominus", "compute_distance", "compute_angle"):
        m = re.search(r"^def %s\(.*?(?=^def |\Z)" % name, src, re.S | re.M)
        if not m:
            raise SystemExit(f"{name} not found in {REF}")
        # (doc strings are expressions of constants: allowed; a trailing comment block belongs to no statement)
        exec(compile(_vet(name, m.group(0)), REF, "exec"), ns)
        digest.update(m.group(0).encode())
    ns["_sha256"] = digest.hexdigest()
    return ns


def pose(rng, max_deg, max_t):
    T = numpy.eye(4)
    axis = rng.normal(size=3); axis /= numpy.linalg.norm(axis)
    T[:3, :3] = Rotation.from_rotvec(axis * numpy.deg2rad(rng.uniform(0, max_deg))).as_matrix()
    T[:3, 3] = rng.uniform(-max_t, max_t, 3)
    return T


def main():
    E = reference_functions()
    rng = numpy.random.default_rng(20260930)
    cases = []
    for k in range(96):
        scale = (180.0, 30.0, 3.0, 0.3)[k % 4]
        A = pose(rng, 180.0, 3.0)
        if k % 12 == 11:
            B = A.copy()                                    # identical poses: both errors 0 (up to the acos of a rounded trace)
        else:
            D = pose(rng, scale, 0.5 * scale / 30.0)        # B = A * D: the metric must return D's angle and |t|
            B = A @ D
        rel = E["ominus"](A, B)
        cases.append({"A": A.reshape(16).tolist(), "B": B.reshape(16).tolist(),
                      "angle": float(E["compute_angle"](rel)), "distance": float(E["compute_distance"](rel))})
    json.dump({"source": "tools/evaluate_rpe.py: ominus, compute_angle, compute_distance (executed from /root/reference, not copied)",
               "executed_source_sha256": E["_sha256"],
               "cases": cases}, open(os.path.join(HERE, "pose_error_golden.json"), "w"), indent=0)
    print(len(cases), "cases; angle range", min(c["angle"] for c in cases), max(c["angle"] for c in cases))


if __name__ == "__main__":
    main()
