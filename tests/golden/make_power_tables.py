"""Extracts the published assembly-power tables the reference's drivers check against (data only) into
tests/golden/assembly_powers.json.  Runs in the build container only (reads /root/reference as text).

  IAEA-2D  : tests/iaea2d/iaea2d.py:479-504 (`data_iaea2D`, 19x19, normalised to 177 fuel assemblies, :418-420)
  KOEBERG  : tests/koeberg2d/koeberg2d.py:553-576 (`data_koeberg2D`, 17x17)
"""
import json
import os
import re

import numpy as np

REF = "/root/reference/tests"


def table(path, name):
    src = open(path).read()
    body = src[src.index(name + " = np.array(["):]
    body = body[:body.index("])") + 2]
    rows = re.findall(r"\[([^\[\]]+)\]", body)
    return [[None if "nan" in v else float(v) for v in r.split(",") if v.strip()] for r in rows]


def norm(path):
    m = re.search(r"self\.Fass = ([0-9.]+) \* self\.Fass / self\.Fass\.sum\(\)", open(path).read())
    return float(m.group(1))


if __name__ == "__main__":
    out = {}
    for key, rel, nm in (("iaea2d", "iaea2d/iaea2d.py", "data_iaea2D"), ("koeberg2d", "koeberg2d/koeberg2d.py", "data_koeberg2D")):
        p = os.path.join(REF, rel)
        t = table(p, nm)
        out[key] = dict(source=f"tests/{rel} ({nm})", normalisation=norm(p), table=t)
        a = np.array([[np.nan if v is None else v for v in r] for r in t])
        print(key, a.shape, "fuel assemblies:", int(np.isfinite(a).sum()), "sum:", np.nansum(a), "norm:", out[key]["normalisation"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "assembly_powers.json"), "w") as f:
        json.dump(out, f)
