"""Pretty-prints a bench.py JSON line from stdin."""
import json, sys
d = json.loads(sys.stdin.read())
c = d["config"]
print(d["value"], d["unit"], "| ms/pair", c["ms_per_pair_per_gpu"], "| one batch in flight", c.get("one_batch_in_flight_ms_per_pair"),
      "| single pair", c.get("single_pair_in_flight_ms"), "| copy GB/s", d.get("roofline", {}).get("device_copy_GBps"))
tot = 0.0
for k, v in d.get("kernels", {}).items():
    print(f"  {k:14s} {v['ms_per_pair']:.4f} ms/pair  (level 0 {v['level0_ms_per_pair']:.4f})  launches/step {v['launches_per_step']}")
    tot += v["ms_per_pair"]
print("  sum of kernels", round(tot, 4), "| roofline", {k: v for k, v in d.get("roofline", {}).items() if k != "note"})
print("  pipeline", d.get("pipeline"))
