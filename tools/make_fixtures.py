"""Generate the small data fixtures under tests/golden/ from the reference's own data files.

Run in the build container (needs /root/reference).  Only DATA is extracted: numeric arrays of
examples/spring_mass_utils/{data.c,x0.txt,lambda0_tree.txt} and the six
examples/random_qp_utils/data0[0-5].json files (inputs + YALMIP/quadprog golden xopt/uopt held
by the reference's own unit test examples/random_qp.c).  No reference source text is copied.
"""
from __future__ import annotations

import json
import re
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def parse_c_arrays(text: str) -> dict:
    out = {}
    for m in re.finditer(r"^(int|double)\s+(\w+)\s*(?:\[\d*\])?\s*=\s*(\{[^}]*\}|[^;]+);", text, flags=re.M):
        typ, name, body = m.groups()
        body = body.strip()
        conv = int if typ == "int" else float
        if body.startswith("{"):
            vals = [conv(v) for v in body.strip("{} \n").replace("\n", " ").split(",") if v.strip()]
            out[name] = vals
        else:
            out[name] = conv(body)
    return out


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    sm = parse_c_arrays((REF / "examples/spring_mass_utils/data.c").read_text())
    sm["x0"] = [float(v) for v in (REF / "examples/spring_mass_utils/x0.txt").read_text().split()]
    sm["lambda0_tree"] = [float(v) for v in (REF / "examples/spring_mass_utils/lambda0_tree.txt").read_text().split()]
    sm["_source"] = "examples/spring_mass_utils/{data.c,x0.txt,lambda0_tree.txt} (numeric data only)"
    (OUT / "spring_mass_data.json").write_text(json.dumps(sm, indent=0))
    # the two text inputs the unchanged spring-mass driver reads at run time (plain data files)
    for name in ("x0.txt", "lambda0_tree.txt"):
        (OUT / name).write_text((REF / "examples/spring_mass_utils" / name).read_text())
    for i in range(6):
        d = json.loads((REF / f"examples/random_qp_utils/data0{i}.json").read_text())
        d["_source"] = f"examples/random_qp_utils/data0{i}.json (reference unit-test fixture incl. golden xopt/uopt)"
        (OUT / f"random_qp_data0{i}.json").write_text(json.dumps(d))
    print("wrote", sorted(p.name for p in OUT.glob("*.json")))


if __name__ == "__main__":
    main()
