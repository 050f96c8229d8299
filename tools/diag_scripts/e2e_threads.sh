# end_to_end of bench.py for several sizes of the host packing pool (gfbe_options.host_threads)
for n in ${THREADS:-32 64 128}; do
  python bench.py --no-cpu-baseline --no-single --mixed 0 --steps 5 --host-threads $n 2>/dev/null > /tmp/e2e_$n.json
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
d = json.load(open("/tmp/e2e_%s.json" % n)); e = d["end_to_end"]
print(n, "threads: host_fed", round(e["host_fed"]["value"]), "solves/s, upload call", round(e["host_fed"]["host_ms_in_upload_call"], 2), "ms, download call", round(e["host_fed"]["host_ms_in_download_call"], 2), "ms; table_fed", round(e["table_fed"]["value"]))
PY
done
