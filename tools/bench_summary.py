"""Print the main figures of a bench.py JSON line (file given as argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"]), "samples/s", round(d["ms_per_step"], 4), "ms/step |", d["dtype"][:60])
for k in ("strict", "fast_wgrad", "full_first_layer", "n68"):
    if k in d:
        print(" ", k, round(d[k]["ms_per_step"], 4), "ms/step", round(d[k]["value"]))
if "script_batch" in d:
    print("  script_batch", {k: round(v["ms_per_step"], 4) for k, v in d["script_batch"].items() if isinstance(v, dict)})
if "exchange_rehearsal" in d:
    print("  exchange_rehearsal", json.dumps(d["exchange_rehearsal"])[:700])
for b, e in d.get("fused_loss_kernels", {}).items():
    if isinstance(e, dict):
        print("  fused_loss", b, {k: round(v["frac"], 3) for k, v in e.items()})
for key in ("roofline", "roofline_fused_loss"):
    if key in d:
        r = d[key]
        print(" ", key, r.get("kernel"), round(r["frac"], 3), r["unit"], round(r["achieved"], 1), "avg_ms", round(r["avg_ms"], 4))


def table(ks, steps):
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["share_of_step"]):
        print(f"    {k:62s} {v['avg_ms'] * v['launches'] / steps:8.4f} ms/step  x{v['launches'] // steps:<3d} frac {v['frac']:.3f}  share {v['share_of_step']:.3f}")


table(d.get("kernels", {}), d["steps"])
for k in ("also", "also_bf16"):
    if k in d:
        e = d[k]
        print(k, round(e["value"]), "samples/s", round(e["ms_per_step"], 3), "ms/step elementwise", e.get("elementwise_share_of_step"))
        table(e.get("kernels", {}), e["steps"])
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"])
