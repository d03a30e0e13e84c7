"""Documentation consistency (CPU): every profile / tool / source file that DESIGN.md, INTEGRATION.md and
profiles/README.md point at exists, and the committed bench line carries the fields the contract asks for."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def referenced_paths(md):
    text = open(os.path.join(ROOT, md)).read()
    pats = re.findall(r"`((?:profiles|tools|tests|oracle|include|faer-rs_amd)/[A-Za-z0-9_./\-]+\.[a-z]+)`", text)
    pats += [f"profiles/{p}" for p in re.findall(r"`(r0[123]_[A-Za-z0-9_/]+\.(?:txt|csv|json))`", text)]
    return sorted(set(pats))


def test_referenced_files_exist():
    missing = []
    for md in ("DESIGN.md", "INTEGRATION.md", "profiles/README.md", "README.md"):
        for p in referenced_paths(md):
            if "*" in p or "{" in p:
                continue
            if not os.path.exists(os.path.join(ROOT, p)):
                missing.append((md, p))
    assert not missing, missing


import pytest


@pytest.mark.parametrize("name", ["r01_final_bench_line.json", "r02_bench_line.json", "r03_bench_line.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    line = json.load(open(os.path.join(ROOT, "profiles", name)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    assert line["config"]["workload"].startswith("dgemm_f64_n8192") and line["cpu_baseline"]["kind"] in ("port", "reference")
