#!/usr/bin/env python
"""Pin the WEG step-size schedule of ``Convofusion._diffusion_reverse`` with the REFERENCE's own statement.

The reference loop re-assigns ``scale_range`` inside the loop (convofusion/models/modeltype/convofusion.py:442-444) and
never resets it, unlike the rollout (unbounded_synthesis.py:82-89).  The module cannot be imported here (pytorch_lightning,
omegaconf ... are missing), so this script parses the reference file, takes the assignment node itself out of the ``for``
loop of ``_diffusion_reverse`` and of ``diffusion_reverse_forecast`` (plus the statements that (re)initialise the name) and
executes exactly those nodes N times.  Only the resulting numbers are stored:  tests/golden/weg_scale_schedule.npz
    carry_N / fresh_N : scale_range[i] used at iteration i, for N in (3, 20, 1000), scale_range = [1.0, 0.5]
Build container only (needs /root/reference).  Usage: python tests/golden/make_golden_weg_schedule.py
"""
import ast
import os
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def loop_assignment(path, func_name):
    """(init statements before the loop, statements inside the `for i, t in enumerate(...)` loop) that assign `scale_range`."""
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == func_name)
    loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For) and isinstance(n.target, ast.Tuple) and n.target.elts[0].id == "i")
    assigns = lambda root: [n for n in ast.walk(root) if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "scale_range" for t in n.targets)]
    inside = sorted(assigns(loop), key=lambda n: n.lineno)
    before = [n for n in assigns(fn) if n.lineno < loop.lineno]
    return before, inside


def run(path, func_name, N, cfg_range):
    before, inside = loop_assignment(path, func_name)
    timesteps = list(range(N))
    env = {"np": np, "self": SimpleNamespace(weg_parameters={"scale_range": list(cfg_range)}, scheduler=SimpleNamespace(timesteps=timesteps)),
           "model": SimpleNamespace(scheduler=SimpleNamespace(timesteps=timesteps))}
    for node in before:
        exec(compile(ast.Module([node], []), path, "exec"), env)
    used = []
    for i in range(N):
        for node in inside:
            exec(compile(ast.Module([node], []), path, "exec"), env)
        used.append(float(env["scale_range"][i]))
    return np.array(used), [n.lineno for n in before + inside]


def main():
    out = {}
    for N in (3, 20, 1000):
        out[f"carry_{N}"], l1 = run("/root/reference/convofusion/models/modeltype/convofusion.py", "_diffusion_reverse", N, (1.0, 0.5))
        out[f"fresh_{N}"], l2 = run("/root/reference/unbounded_synthesis.py", "diffusion_reverse_forecast", N, (1.0, 0.5))
    print("statements executed: convofusion.py lines", l1, "; unbounded_synthesis.py lines", l2)
    print("N=1000: carry[1] =", out["carry_1000"][1], " carry[400] =", out["carry_1000"][400], " fresh[400] =", out["fresh_1000"][400])
    np.savez_compressed(os.path.join(HERE, "weg_scale_schedule.npz"), **out)


if __name__ == "__main__":
    main()
