import os, sys, subprocess, ctypes as C
import numpy as np
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tests')]
from _gfbe_import import gf
import importlib.util
spec = importlib.util.spec_from_file_location("dump_stream", "tools/dump_stream.py"); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
abi, stream = gf.abi, gf.stream
S = stream.Stream(seed=3, n_kf=28, new_per_frame=50)
mod.dump("/tmp/s.bin", S)
libdir=os.path.dirname(gf.lib_path())
subprocess.run(["g++","-O2","-std=c++17","-I","include","examples/stream_loop.cpp","-L",libdir,"-lgfbe","-Wl,-rpath,"+libdir,"-o","/tmp/sl"],check=True)
print(subprocess.run(["/tmp/sl","/tmp/s.bin","/tmp/t.bin"],capture_output=True,text=True).stdout)
raw=open("/tmp/t.bin","rb").read(); n=int(np.frombuffer(raw[:4],np.int32)[0])
traj=np.frombuffer(raw[4:4+56*n]).reshape(n,7); costs=np.frombuffer(raw[4+56*n:4+64*n])
ints=np.frombuffer(raw[4+64*n:],np.int32).reshape(4,n)
be = gf.Backend(device=0)
T = abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 16384, options=dict(min_parallax=14.0 / 600, depth_threshold=6.0))
# count outliers in python loop
cnt=[]
orig=T.check_outliers
def co(*a,**k):
    r=orig(*a,**k); cnt.append(len(r[0])); return r
T.check_outliers=co
slide = be.lib.gfbe_slide_window_state
ref = stream.run_stream(be, T, S, lambda st, flag: slide(C.byref(st), int(flag)), device_handoff=True)
for i in range(n):
    print(i, ints[1][i], ref["flags"][i], ints[0][i], ref["iterations"][i], "L", ints[2][i], ref["n_landmarks"][i], "out", ints[3][i], cnt[i], "cost rel %.2e"%(abs(costs[i]-ref["final_cost"][i])/costs[i]), "traj %.2e"%np.abs(traj[i]-ref["traj"][i]).max())
