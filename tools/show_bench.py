"""Pretty-print bench.py JSON lines (per-kernel table)."""
import json
import sys

for f in sys.argv[1:]:
    print("==", f)
    for l in open(f):
        l = l.strip()
        if not l.startswith("{"):
            continue
        d = json.loads(l)
        print(f"  value {d['value']} img/s   ms/step {d['ms_per_step']}   step: {d['step_roofline']}")
        r = d["roofline"]
        print(f"  dominant {r['kernel']} {r['achieved']} GB/s frac {r['frac']} ({r['us']} us)")
        print("  " + "  ".join(f"{k.split('.')[0][0]}.{k.split('.')[1]}={v['us']}us" + (f"({v['GBps']:.0f}GB/s)" if 'GBps' in v else "") for k, v in d["kernels"].items()))
        if d.get("cpu_baseline"):
            print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "cores")
