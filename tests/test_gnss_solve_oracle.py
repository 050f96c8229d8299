"""GNSS inside the window solve and the marginalisation, CPU oracle (estimator.cpp:2965-3002, 3239-3291, 3462-3496, 3561-3590).
The wiring — which tangent dims each GNSS block lands on, the lower_idx / ts_ratio interpolation, the lowspeed gate, the drop
sets and the receiver-clock address shift — is checked against numpy assemblies built from the stand-alone factor evaluator
(`gfo_gnss_eval`, pinned on its own in tests/test_gnss_oracle.py) and against the independent trust-region loop."""
import numpy as np
import pytest

from _gfbe_import import gf
import ceres_trust_region_np as tr
import gnss_window_cases as gw

abi = gf.abi
TOFF = abi.block_tangent_offset


def gnss_rows(oracle, snap, only_frame0=False):
    """[(r, J, [(block id, columns of J)])] of the GNSS residual blocks, from the stand-alone evaluator."""
    gn, gs = snap["gnss"], snap["gnss_state"]
    e = abi.gnss_eval(oracle.lib, "gfo_", None, gn["obs"], gn.get("iono"), snap["pose"], snap["speed_bias"], gs["rcv_dt"], gs["rcv_ddt"], gs["yaw_enu_local"],
                      gs["anc_ecef"], gn["frame_dt"], gn["ddt_weight"])
    rows = []
    for k, o in enumerate(gn["obs"]):
        lo, fr = int(o["lower_idx"]), int(o["frame"])
        if only_frame0 and fr != 0:
            continue
        J = e["J"][k]
        rows.append((e["r"][k], J, [(abi.BLK_POSE0 + lo, [0, 1, 2]), (abi.BLK_SB0 + lo, [0, 1, 2]), (abi.BLK_POSE0 + lo + 1, [0, 1, 2]), (abi.BLK_SB0 + lo + 1, [0, 1, 2]),
                                    (abi.BLK_RCV_DT0 + 4 * fr + int(o["sys_idx"]), [0]), (abi.BLK_RCV_DDT0 + fr, [0]), (abi.BLK_YAW_ENU, [0]), (abi.BLK_ANC_ECEF, [0, 1, 2])]))
    fdt, wgt = np.asarray(gn["frame_dt"], float), float(gn["ddt_weight"])
    for k in range(4):
        for i in range(1 if only_frame0 else abi.WINDOW_SIZE):
            rows.append((e["r_dt_ddt"][k, i:i + 1], np.array([[-50.0, 50.0, -25.0 * fdt[i], -25.0 * fdt[i]]]),
                         [(abi.BLK_RCV_DT0 + 4 * i + k, [0]), (abi.BLK_RCV_DT0 + 4 * (i + 1) + k, [0]), (abi.BLK_RCV_DDT0 + i, [0]), (abi.BLK_RCV_DDT0 + i + 1, [0])]))
    for i in range(1 if only_frame0 else abi.WINDOW_SIZE):
        rows.append((e["r_smooth"][i:i + 1], np.array([[wgt, -wgt]]), [(abi.BLK_RCV_DDT0 + i, [0]), (abi.BLK_RCV_DDT0 + i + 1, [0])]))
    return rows


def test_gnss_part_of_the_normal_equations(oracle):
    _, _, snap = gw.gnss_window(seed=83, L=60, n_per_frame=5)
    base = dict(snap)
    base.pop("gnss")
    a, b = oracle.linearize(snap), oracle.linearize(base)
    H, g, cost = np.zeros_like(a["H"]), np.zeros_like(a["g"]), 0.0
    for r, J, blocks in gnss_rows(oracle, snap):
        cols = np.concatenate([TOFF(bid) + np.asarray(c) for bid, c in blocks])
        H[np.ix_(cols, cols)] += J.T @ J
        g[cols] += J.T @ r
        cost += 0.5 * float(r @ r)
    yaw = TOFF(abi.BLK_YAW_ENU)      # held constant (estimator.cpp:2991): not in the reduced program
    H[yaw, :] = 0.0
    H[:, yaw] = 0.0
    g[yaw] = 0.0
    dH, dg = a["H"] - b["H"], a["g"] - b["g"]
    assert np.abs(dH - H).max() < 1e-12 * np.abs(a["H"]).max()
    assert np.abs(dg - g).max() < 1e-11 * np.abs(a["g"]).max()
    assert abs((a["cost"] - b["cost"]) - cost) < 1e-10 * a["cost"]
    # the receiver clock, the anchor and the poses / velocities the observations interpolate between are coupled
    o = snap["gnss"]["obs"][0]
    assert H[TOFF(abi.BLK_RCV_DT0 + 4 * o["frame"] + o["sys_idx"]), TOFF(abi.BLK_POSE0 + o["lower_idx"])] != 0.0
    assert np.abs(H[TOFF(abi.BLK_ANC_ECEF):TOFF(abi.BLK_ANC_ECEF) + 3, TOFF(abi.BLK_RCV_DDT0):]).max() == 0.0      # (the Doppler row has no anchor column)
    assert np.abs(a["H"][:, yaw]).max() == 0.0 and np.abs(b["H"][TOFF(abi.BLK_ANC_ECEF):, :]).max() == 0.0


@pytest.mark.parametrize("seed,anchor", [(81, False), (85, True)])
def test_gnss_solve_matches_the_independent_loop(oracle, seed, anchor):
    _, _, snap = gw.gnss_window(seed=seed, L=80, n_per_frame=6, anchor=anchor)
    res = oracle.solve(snap, abi.MARGIN_NONE)
    ref = tr.solve(oracle, snap)
    s = res["summary"]
    assert s["iterations"] == ref["iterations"] and s["termination"] == ref["termination"]
    assert list(s["accepted"]) == ref["accepted"]
    np.testing.assert_allclose(s["cost_history"], ref["cost_history"], rtol=1e-7)
    g, gr = res["state"]["gnss_state"], ref["snap"]["gnss_state"]
    np.testing.assert_allclose(g["rcv_dt"], gr["rcv_dt"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(g["rcv_ddt"], gr["rcv_ddt"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(g["anc_ecef"], gr["anc_ecef"], rtol=0, atol=1e-5)
    assert g["yaw_enu_local"] == snap["gnss_state"]["yaw_enu_local"]          # constant
    assert s["final_cost"] < 1e-3 * s["initial_cost"] and s["iterations"] >= 3
    assert np.abs(g["rcv_dt"] - snap["gnss_state"]["rcv_dt"]).max() > 1e-2   # the clock moved


def test_slow_window_has_no_gnss_factors_but_still_marginalises_them(oracle):
    _, _, snap = gw.gnss_window(seed=86, L=60, n_per_frame=4)
    slow = dict(snap)
    slow["speed_bias"] = np.array(snap["speed_bias"], float).copy()
    slow["speed_bias"][:, :2] *= 0.2                                   # mean |v_xy| ~ 0.2 m/s < 0.3: lowspeed
    assert not tr.gnss_factors_on(slow) and tr.gnss_factors_on(snap)
    off = dict(slow)
    off.pop("gnss")
    a, b = oracle.solve(slow, abi.MARGIN_OLD), oracle.solve(off, abi.MARGIN_OLD)
    assert a["summary"]["cost_history"] == b["summary"]["cost_history"]
    assert a["state"]["gnss_state"]["rcv_dt"].tolist() == np.asarray(snap["gnss_state"]["rcv_dt"]).tolist()
    ids = a["prior"]["block_id"].tolist()
    assert abi.BLK_ANC_ECEF in ids and abi.BLK_YAW_ENU in ids and abi.BLK_RCV_DDT0 in ids and abi.BLK_ANC_ECEF not in b["prior"]["block_id"].tolist()


def numpy_marginalize(oracle, snap, rows_extra, drop_extra):
    """Dense Schur complement over blocks in the oracle's canonical order (dropped blocks by id, landmarks, kept blocks by id)."""
    ev = oracle.eval_factors(snap, robustify=True)
    sel = np.where(np.asarray(snap["vis_imu_i"]) == 0)[0]
    lms = sorted(set(np.asarray(snap["vis_feature_index"])[sel].tolist()))
    rows = []
    rows.append((ev["imu_r"][0], ev["imu_J"][0], [(0, range(6)), (11, range(9)), (1, range(6)), (12, range(9))]))
    rows.append((ev["wheel_r"][0], ev["wheel_J"][0], [(0, range(6)), (1, range(6)), (23, range(6)), (24, [0]), (25, [0]), (26, [0]), (28, [0])]))
    for k in sel:
        j, l = int(snap["vis_imu_j"][k]), int(snap["vis_feature_index"][k])
        rows.append((ev["vis_r"][k], ev["vis_J"][k], [(0, range(6)), (j, range(6)), (22, range(6)), (("l", l), [0]), (27, [0])]))
    rows += rows_extra
    pr = snap.get("prior")
    if pr is not None:
        n = int(pr["n"])
        blocks = [(int(pr["block_id"][q]), range(abi.block_local_size(int(pr["block_id"][q])))) for q in np.argsort(pr["block_idx"])]
        rows.append((ev["prior_r"], np.asarray(pr["J0"]).reshape(n, n), blocks))
    touched = set()
    for _, _, blocks in rows:
        touched |= {b for b, _ in blocks if not isinstance(b, tuple)}
    drop = [b for b in sorted(touched) if b in (0, 11) or b in drop_extra]
    keep = [b for b in sorted(touched) if b not in drop]
    idx, pos = {}, 0
    for b in drop:
        idx[b] = pos
        pos += abi.block_local_size(b)
    for l in lms:
        idx[("l", l)] = pos
        pos += 1
    m = pos
    for b in keep:
        idx[b] = pos
        pos += abi.block_local_size(b)
    A, bb = np.zeros((pos, pos)), np.zeros(pos)
    for r, J, blocks in rows:
        cols = np.concatenate([idx[b] + np.asarray(list(c)) for b, c in blocks])
        A[np.ix_(cols, cols)] += J.T @ J
        bb[cols] += J.T @ r
    w, V = np.linalg.eigh(0.5 * (A[:m, :m] + A[:m, :m].T))
    Ainv = (V * np.where(w > 1e-8, 1.0 / w, 0.0)) @ V.T
    return keep, A[m:, m:] - A[m:, :m] @ Ainv @ A[:m, m:], bb[m:] - A[m:, :m] @ Ainv @ bb[:m], np.abs(A).max(), np.abs(bb).max()


def shifted_old(ids):
    out = []
    for b in ids:
        if b < abi.BLK_EX_CAM or b >= abi.BLK_RCV_DDT0:
            out.append(b - 1)
        elif b >= abi.BLK_RCV_DT0:
            out.append(b - 4)
        else:
            out.append(b)
    return out


def test_marginalise_old_with_gnss_against_numpy_schur(oracle):
    scn, tru, snap = gw.gnss_window(seed=87, L=80, n_per_frame=6)
    drop_extra = [abi.BLK_RCV_DT0 + k for k in range(4)] + [abi.BLK_RCV_DDT0]
    keep, Ap, bp, a_scale, b_scale = numpy_marginalize(oracle, snap, gnss_rows(oracle, snap, only_frame0=True), drop_extra)
    prior, A, b, rc = oracle.marginalize(snap, abi.MARGIN_OLD)
    assert rc == 0 and prior["valid"] == 1
    assert prior["block_id"].tolist() == shifted_old(keep)
    # kept GNSS blocks: rcv_dt[1][k] -> rcv_dt[0][k], rcv_ddt[1] -> rcv_ddt[0], the yaw (a kept parameter here) and the anchor
    for bid in [abi.BLK_RCV_DT0 + k for k in range(4)] + [abi.BLK_RCV_DDT0, abi.BLK_YAW_ENU, abi.BLK_ANC_ECEF]:
        assert bid in prior["block_id"].tolist()
    assert prior["n"] == Ap.shape[0] == sum(abi.block_local_size(q) for q in keep)
    assert np.abs(A - Ap).max() < 1e-11 * a_scale
    assert np.abs(b - bp).max() < 1e-11 * max(b_scale, a_scale * 1e-2)
    J0, r0 = prior["J0"], prior["r0"]
    w, V = np.linalg.eigh(0.5 * (Ap + Ap.T))
    k = w > 1e-8
    assert np.abs(J0.T @ J0 - (V[:, k] * w[k]) @ V[:, k].T).max() < 1e-11 * a_scale
    # x0: the clock values of frame 1, the yaw and the anchor at marginalisation time
    x0, off = prior["x0"], 0
    gs = snap["gnss_state"]
    for bid, size in zip(prior["block_id"].tolist(), prior["block_size"].tolist()):
        if abi.BLK_RCV_DT0 <= bid < abi.BLK_RCV_DDT0:
            assert x0[off] == np.asarray(gs["rcv_dt"])[1, bid - abi.BLK_RCV_DT0]
        elif bid == abi.BLK_RCV_DDT0:
            assert x0[off] == np.asarray(gs["rcv_ddt"])[1]
        elif bid == abi.BLK_YAW_ENU:
            assert x0[off] == gs["yaw_enu_local"]
        elif bid == abi.BLK_ANC_ECEF:
            assert x0[off:off + 3].tolist() == np.asarray(gs["anc_ecef"]).tolist() and size == 3
        off += size


def test_gnss_prior_chain(oracle):
    """window 0 --MARGIN_OLD--> prior with the clock / yaw / anchor blocks --> window 1: the prior's residual is r0 for the blocks
    that did not move; MARGIN_SECOND_NEW keeps them where they are; MARGIN_OLD again drops rcv_dt[0], rcv_ddt[0] through the prior
    AND the new frame-0 factors (numpy Schur complement with the prior as one more residual block)."""
    scn, tru, snap = gw.gnss_window(seed=88, L=300, n_per_frame=5)
    resA = oracle.solve(snap, abi.MARGIN_OLD)
    nxt = gw.next_gnss_window(scn, tru, resA, seed=88)
    ev = oracle.eval_factors(nxt)
    assert np.abs(ev["prior_r"] - resA["prior"]["r0"]).max() < 1e-6 * max(1.0, np.abs(resA["prior"]["r0"]).max())
    resB = oracle.solve(nxt, abi.MARGIN_SECOND_NEW)
    ids_a, ids_b = resA["prior"]["block_id"].tolist(), resB["prior"]["block_id"].tolist()
    assert ids_b == [b for b in ids_a if b != abi.BLK_POSE0 + 9] and resB["prior"]["n"] == resA["prior"]["n"] - 6
    ref = tr.solve(oracle, nxt)
    assert list(resB["summary"]["accepted"]) == ref["accepted"]
    np.testing.assert_allclose(resB["summary"]["cost_history"], ref["cost_history"], rtol=1e-7)
    drop_extra = [abi.BLK_RCV_DT0 + k for k in range(4)] + [abi.BLK_RCV_DDT0]
    keep, Ap, bp, a_scale, b_scale = numpy_marginalize(oracle, nxt, gnss_rows(oracle, nxt, only_frame0=True), drop_extra)
    prior, A, b, rc = oracle.marginalize(nxt, abi.MARGIN_OLD)
    assert rc == 0 and prior["block_id"].tolist() == shifted_old(keep)
    assert np.abs(A - Ap).max() < 1e-10 * a_scale
    assert np.abs(b - bp).max() < 1e-10 * max(b_scale, a_scale * 1e-2)
