cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/full_tests.log 2>&1
(timeout 600 python bench.py 2> gpurun_out/full_bench.err | tail -1) > gpurun_out/full_bench.json
tail -4 gpurun_out/full_tests.log
python - <<'P'
import json
d=json.load(open("gpurun_out/full_bench.json"))
print("value", d["value"], "resident_1024", d["resident_1024"]["value"], "mixed", d["mixed_batch"]["value"] if d.get("mixed_batch") else None)
print("single_ms", d["single_window_ms"], "h2h", d["single_window_host_to_host_ms"])
print("e2e", {k: v["value"] for k, v in d["end_to_end"].items() if isinstance(v, dict) and "value" in v})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("reference_construction", {}).get("median_ms"), d["cpu_baseline"].get("product_algorithm", {}).get("median_ms"))
print("like", json.dumps(d.get("speedup_like_for_like"))[:900])
print("roofline frac", d["roofline"]["frac"], "on counters", d["roofline"].get("frac_on_counter_bytes"), "kernels", {k: (round(v.get("avg_launch_us"), 1), round(v.get("frac"), 3), v.get("frac_of_hbm_on_counter_bytes")) for k, v in d["roofline"].get("kernels", {}).items()})
print("other", json.dumps(d.get("other_configs"))[:3000])
print("cpu per-window", d["cpu_baseline"]["reference_construction"].get("per_window_median_ms"), d["cpu_baseline"]["reference_construction"].get("p10_ms"), d["cpu_baseline"]["reference_construction"].get("p90_ms"))
P
