"""stdin: the JSON line of bench.py; prints the headline and the `ms` of every entry of `others` (A/B runs of bench options)"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
o = d.get("others", {})
print(sys.argv[1] if len(sys.argv) > 1 else "", "headline", d["value"], {k: o[k]["ms"] for k in o if "ms" in o[k]})
