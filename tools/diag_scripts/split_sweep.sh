# resident throughput of the default workload for several numbers of parts (gfbe_options.split_batch)
for sp in ${SPLITS:-4 2 6 8}; do
  python bench.py --no-cpu-baseline --no-e2e --no-single --mixed 0 --steps 10 --split $sp 2>/dev/null > /tmp/split_$sp.json
  python - $sp <<'PY'
import json, sys
d = json.load(open("/tmp/split_%s.json" % sys.argv[1]))
print("split", sys.argv[1], ":", round(d["value"]), "solves/s at 8192 windows,", round(d["resident_1024"]["value"]), "at 1024")
PY
done
