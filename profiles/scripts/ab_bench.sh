#!/bin/bash
cd /root/repo
P='import json,sys; a=json.loads(sys.stdin.readline()); e=a["end_to_end"]; print("value", round(a["value"]), "r1024", round(a["resident_1024"]["value"]), "single", round(a["single_window_ms"],4), "e2e", round(e["host_fed"]["value"]), round(e["table_fed"]["value"]))'
for rep in 1 2; do
echo "== base (start of the session)"; GFBE_LIB=/root/repo/ground-fusion2_amd/csrc/variants/libgfbe_r3base.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== now"; python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
done
