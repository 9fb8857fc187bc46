import ctypes, sys
sys.path.insert(0, '.')
from flatland_marl_amd import workload as wl, hip_backend as hb
from flatland_marl_amd.hip_backend import BatchedRailEnv
for w in ("cfg1","cfg2","cfg3","cfg4","cfg5"):
    envs,_=wl.make_envs(w,B=1)
    env=BatchedRailEnv(envs)
    L=hb.lib()
    L.fl_debug_obs_config.argtypes=[ctypes.c_void_p,ctypes.c_int,ctypes.c_int,ctypes.c_int,ctypes.c_void_p]
    for depth in (2,3):
        out=(ctypes.c_int*11)()
        rc=L.fl_debug_obs_config(env.h,500,depth,30,out)
        print(w, "depth", depth, rc, dict(nt=out[0],lds=out[1],tab=out[2],nh=out[3],wl=out[4],tmask=out[5],dual=out[6],items=out[7],merged=out[8],compact=out[9],fix=out[10]))
