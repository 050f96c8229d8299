"""Writes a synthetic keyframe stream (ground-fusion2_amd/stream.py::Stream) to a flat binary file for the compiled loop driver
examples/stream_loop.cpp: tracker output per keyframe, raw IMU / wheel samples per interval, the initial window state, the
extrinsics and the noise parameters. Little-endian, int32 and float64 only, in the order read_stream() of the driver reads them.

    python tools/dump_stream.py out.bin [seed n_kf new_per_frame]
"""
import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
import numpy as np

MAGIC = 0x47465354


def dump(path, S, min_parallax=14.0 / 600, depth_threshold=6.0):
    from _gfbe_import import gf
    abi, synth = gf.abi, gf.synth
    scn = S.scn
    W = abi.WINDOW_SIZE
    st = scn.initial_state(0)
    with open(path, "wb") as f:
        def I(*v):
            f.write(np.asarray(v, np.int32).tobytes())

        def D(v):
            f.write(np.ascontiguousarray(np.asarray(v, np.float64)).tobytes())
        I(MAGIC, S.n_kf, W, int(S.use_wheel))
        D([min_parallax, depth_threshold])
        D(scn.tic); D(scn.ric)
        D(scn.ba_est); D(scn.bg_est)
        D([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W]); D([synth.VEL_N_WHEEL, synth.GYR_N_WHEEL])
        D(st["pose"]); D(st["speed_bias"]); D(st["ex_pose"]); D(st["ex_pose_wheel"]); D(st["ix_wheel"]); D([st["td"], st["td_wheel"]])
        for ids, obs in S.frames:
            I(len(ids)); I(*ids); D(obs)
        for raw in ([scn.imu_raw] + ([scn.wheel_raw] if S.use_wheel else [])):
            I(len(raw))
            for samples, first in raw:
                I(len(samples)); D(samples); D(first)


if __name__ == "__main__":
    from _gfbe_import import gf
    out = _sys.argv[1]
    seed, n_kf, new = (int(v) for v in (_sys.argv[2:5] + ["3", "36", "250"][len(_sys.argv) - 2:]))
    dump(out, gf.stream.Stream(seed=seed, n_kf=n_kf, new_per_frame=new))
    print("wrote", out)
