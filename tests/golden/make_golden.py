"""Extracts the reference's own known-answer vectors for the hot path into
JSON fixtures (run in the build container only; /root/reference is not
available on the GPU box).

Sources (faer 0.24.4):
  * faer/src/linalg/qr/mod.rs:116-191  `test_example`  -- 10x2 least squares
    problem with the numpy solution, tolerance 1e-6 (the only true KAT on
    the path);
  * faer/src/linalg/matmul/mod.rs:1595-1614 -- 2x2 matmul doctest, alpha=2.5,
    tolerance 1e-10.
"""
import json
import os
import re

REF = "/root/reference/faer/src/linalg"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_mats(text):
    """returns every `mat![ [..], [..] ]` literal as a list of rows of floats"""
    out = []
    for m in re.finditer(r"mat!\[(.*?)\];", text, flags=re.S):
        body = m.group(1)
        rows = re.findall(r"\[([^\[\]]*)\]", body)
        mat = []
        ok = True
        for r in rows:
            vals = [v.strip() for v in r.split(",") if v.strip()]
            try:
                mat.append([float(v.replace("_f64", "")) for v in vals])
            except ValueError:
                ok = False
                break
        if ok and mat:
            out.append(mat)
    return out


def main():
    src = open(os.path.join(REF, "qr/mod.rs")).read()
    test = src[src.index("fn test_example"):]
    a, b, x = parse_mats(test)[:3]
    assert len(a) == 10 and len(a[0]) == 2 and len(b) == 10 and len(b[0]) == 3 and len(x) == 2
    json.dump({"source": "faer/src/linalg/qr/mod.rs:116-191", "tol": 1e-6, "a": a, "b": b, "expected_solution": x},
              open(os.path.join(HERE, "qr_lstsq_10x2.json"), "w"), indent=1)

    src = open(os.path.join(REF, "matmul/mod.rs")).read()
    doc = src[src.index("/// # Example"):]
    doc = "\n".join(l.lstrip("/ ").rstrip() for l in doc.splitlines()[:30])
    lhs, rhs = parse_mats(doc)[:2]
    assert lhs == [[0.0, 2.0], [1.0, 3.0]] and rhs == [[4.0, 6.0], [5.0, 7.0]]
    target = [[2.5 * sum(lhs[i][k] * rhs[k][j] for k in range(2)) for j in range(2)] for i in range(2)]
    json.dump({"source": "faer/src/linalg/matmul/mod.rs:1595-1614", "tol": 1e-10, "alpha": 2.5, "lhs": lhs,
               "rhs": rhs, "target": target},
              open(os.path.join(HERE, "matmul_2x2.json"), "w"), indent=1)
    print("wrote qr_lstsq_10x2.json, matmul_2x2.json")


if __name__ == "__main__":
    main()
