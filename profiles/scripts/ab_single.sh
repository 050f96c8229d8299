#!/bin/bash
cd /root/repo
for v in "$@"; do echo "== $v"; GFBE_LIB=/root/repo/ground-fusion2_amd/csrc/variants/libgfbe_$v.so timeout 300 python tools/diag_scripts/single_ms.py 2>&1 | tail -5 | head -2; done
echo "== default"; timeout 300 python tools/diag_scripts/single_ms.py 2>&1 | tail -5 | head -2
