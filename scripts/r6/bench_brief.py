"""One line of the essentials of a bench.py JSON: python scripts/r6/bench_brief.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("C4 %.2f M pairs/s (%.3f ms/step) | solo %.3f ms | two-wave solo %s | c5 %s ms (%.2f M) | e2e %.2f M | replan %.3f M staged / %s pipelined | single %.3f ms" % (
    d["value"] / 1e6, d["ms_per_step"], r.get("solo", {}).get("step_ms_median", float("nan")),
    r.get("solo_two_wavefronts_per_simd", {}).get("step_ms_median"), round(d.get("c5", {}).get("step_ms_median", float("nan")), 2),
    d.get("c5", {}).get("pairs_per_s", float("nan")) / 1e6, (d.get("e2e_with_copies", {}).get("process_with_default_hardware_queues") or d.get("e2e_with_copies", {})).get("pairs_per_s", float("nan")) / 1e6,
    d.get("replan_faithful", {}).get("replans_per_s", float("nan")) / 1e6,
    (d.get("replan_faithful", {}).get("pipelined") or {}).get("replans_per_s"), d.get("single_replan_latency_ms", {}).get("median_ms", float("nan"))))
