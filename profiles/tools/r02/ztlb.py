import sys; import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np
from helpers import *
for shape in [(512,512,128),(128,128,512),(256,256,512),(512,512,256),(512,256,512)]:
    inp = synthetic_inputs(*shape, ng=1, seed=1, nonuniform=False)
    s = make_hip(inp); s.profile_reset()
    ms = s.time_schur_apply(0, 10)
    N = s.ne
    p = {nm: (lambda c,m: m/max(c,1))(*s.profile(nm)) for nm in ("schur_x","schur_y","schur_z")}
    print(shape, "cells %.0fM" % (N/1e6), "x %.3f y %.3f z %.3f ms" % (p['schur_x'], p['schur_y'], p['schur_z']), "-> GB/s actual(40B/cell): x %.0f y %.0f z %.0f" % tuple(40*N/(p[k]*1e-3)/1e9 for k in ("schur_x","schur_y","schur_z")), flush=True)
    s.close()
