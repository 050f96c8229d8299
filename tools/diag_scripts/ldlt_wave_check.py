import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""k_marg_ldlt_tp (one wave per window, the lower triangle as 9 x 9 register tiles) against k_marg_ldlt<4> (eight waves, all of A'). A
build with -DGFBE_LDLT_TP=2 (tools/diag_variants.py ldlttp2) runs the wave kernel for EVERY batch; single windows — whose other kernels are
the same in both libraries — are solved with both and their new priors compared: byte for byte (they are NOT the same bits: the workgroup
form updates entry (i, j) with (c_i / d) c_j and entry (j, i) with (c_j / d) c_i — its matrix is symmetric to rounding only — and publishes
row p from its upper triangle; the wave form keeps the lower one), and as what the next window uses: J0^T J0 and J0^T r0.
  GFBE_TP2_LIB=ground-fusion2_amd/csrc/variants/libgfbe_ldlttp2.so python tools/diag_scripts/ldlt_wave_check.py"""
import os, numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
ref = gf.Backend(0)
tp2 = gf.Backend(0, so=os.environ["GFBE_TP2_LIB"])
n_cmp = n_same = 0
worst = []
for seed in range(int(os.environ.get("N", "40"))):
    scn = synth.Scenario(seed=500 + seed, n_landmarks=[150, 400, 900, 2000][seed % 4], use_wheel=bool(seed % 3))
    snap = scn.window(0)
    for step in range(2):          # the first window (no prior), then the one that carries its prior on
        flag = abi.MARGIN_OLD if (seed + step) % 4 else abi.MARGIN_SECOND_NEW
        a, b = ref.solve(snap, flag), tp2.solve(snap, flag)
        pa, pb = a["prior"], b["prior"]
        assert (pa is None) == (pb is None)
        if pa is not None:
            n_cmp += 1
            same = (pa["J0"].tobytes() == pb["J0"].tobytes() and pa["r0"].tobytes() == pb["r0"].tobytes() and
                    pa["block_id"].tolist() == pb["block_id"].tolist() and int(pa["n"]) == int(pb["n"]))
            n_same += same
            n = int(pa["n"])
            Ja, Jb = np.asarray(pa["J0"]).reshape(-1)[: n * n].reshape(n, n), np.asarray(pb["J0"]).reshape(-1)[: n * n].reshape(n, n)
            Ha, Hb = Ja.T @ Ja, Jb.T @ Jb
            ga, gb = Ja.T @ np.asarray(pa["r0"])[:n], Jb.T @ np.asarray(pb["r0"])[:n]
            worst.append((float(np.abs(Ha - Hb).max() / np.abs(Ha).max()), float(np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-300)),
                          int((np.abs(Ja).sum(axis=1) > 0).sum()) - int((np.abs(Jb).sum(axis=1) > 0).sum()), float(np.abs(Ja - Jb).max())))
        assert a["state"]["pose"].tobytes() == b["state"]["pose"].tobytes()      # (the solve itself does not involve the kernel)
        if pa is None or step == 1:
            break
        snap = scn.window(step + 1, state=synth.shift_state_for_next_window(scn, a["state"], step + 1), prior=pa)
w = np.array(worst)
print("priors compared: %d, byte for byte the same: %d; largest differences: J0^T J0 %.2e of its largest entry, J0^T r0 %.2e, entries of J0 %.2e; ranks differ in %d" %
      (n_cmp, n_same, w[:, 0].max(), w[:, 1].max(), w[:, 3].max(), int((w[:, 2] != 0).sum())))
