#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Generates tests/golden/rollout_vectors.npz by RUNNING the reference's own
`batch_obs` (ss_baselines/common/utils.py:126-153) and `RolloutStorage`
(ss_baselines/common/rollout_storage.py:14-243) from /root/reference on a seeded scenario.  Only torch
is needed by those two; the modules are loaded by file path with the unrelated imports of utils.py
stubbed out.  The scenario itself lives in tests/rollout_scenario.py and is shared with the parity test,
which replays it through ss_amd.rollout and compares tensor by tensor.

usage: python tests/golden/make_golden_rollout.py      (needs /root/reference; not needed at test time)"""
import ast, importlib.util, os, sys, types
import numpy as np, torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
REF = os.environ.get("SS_REFERENCE", "/root/reference")
import rollout_scenario as S


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_functions(rel, names):
    """exec only the named top-level functions of a reference module (its imports need habitat etc.)"""
    src = open(os.path.join(REF, rel)).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    from collections import defaultdict
    from typing import Dict, List, Optional
    ns = {"torch": torch, "np": np, "defaultdict": defaultdict, "List": List, "Dict": Dict, "Optional": Optional}
    exec(compile(ast.Module(body=keep, type_ignores=[]), rel, "exec"), ns)
    return types.SimpleNamespace(**{n: ns[n] for n in names})


def main():
    ref_rs = load_by_path("ref_rollout_storage", "ss_baselines/common/rollout_storage.py").RolloutStorage
    ref_u = load_functions("ss_baselines/common/utils.py", ["to_tensor", "batch_obs"])
    out = {}
    for sc in S.SCENARIOS:
        res = S.replay(sc, ref_rs, ref_u.batch_obs, device=None, reference=True)
        for k, v in res.items():
            out[f"{sc['name']}/{k}"] = v
    path = os.path.join(HERE, "rollout_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
