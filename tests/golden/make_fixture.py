"""Extracts the only corridor fixture the reference holds for the solver path — the hard-coded
3-polytope corridor and x0/xf of faster/other/gurobi_continuous.cpp:220-222,310-401 — into
tests/golden/fixture_corridor.json.  Inputs only: the reference records no expected outputs.

Run in the build container (needs /root/reference):  python tests/golden/make_fixture.py
"""
import json
import os
import re

SRC = "/root/reference/faster/other/gurobi_continuous.cpp"
HERE = os.path.dirname(os.path.abspath(__file__))


def grab(text, name, start=0):
    m = re.compile(r"\b%s\s*<<" % name).search(text, start)
    end = text.index(";", m.end())
    body = re.sub(r"/+", " ", text[m.end():end])
    nums = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]
    return nums, end


def main():
    text = open(SRC).read()
    out = {"source": "faster/other/gurobi_continuous.cpp:220-222,310-401", "polytopes": []}
    x0 = re.search(r"x0\s*=\s*\{([^}]*)\}", text).group(1)
    xf = re.search(r"xf\s*=\s*\{([^}]*)\}", text).group(1)
    out["x0"] = [float(v) for v in x0.split(",")]
    out["xf"] = [float(v) for v in xf.split(",")]
    for name_a, name_b in (("A1", "b1"), ("A2", "b2"), ("A3", "b3")):
        a, _ = grab(text, name_a)
        b, _ = grab(text, name_b)
        assert len(a) == 3 * len(b), (name_a, len(a), len(b))
        out["polytopes"].append({"A": [a[i:i + 3] for i in range(0, len(a), 3)], "b": b})
    with open(os.path.join(HERE, "fixture_corridor.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()},
          [len(p["b"]) for p in out["polytopes"]])


if __name__ == "__main__":
    main()
