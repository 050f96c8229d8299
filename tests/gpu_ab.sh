R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" prev; do
  if [ -n "$v" ]; then export GFBE_LIB=$R/ground-fusion2_amd/csrc/variants/libgfbe_$v.so; else unset GFBE_LIB; fi
  echo "== ${v:-current} rep $rep"; python - <<'PY'
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
firsts = be.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
for B in (1024, 1):
    b = be.batch_upload([snaps[i % 8] for i in range(B)])
    for _ in range(3): b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize(); n = 10 if B > 1 else 30; t = time.perf_counter()
    for _ in range(n): b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("B=%d: %.3f ms per step, %.0f solves/s" % (B, dt * 1e3, B / dt))
    b.free()
PY
done; done
