import sys, numpy as np
sys.path.insert(0, '/root/repo')
from atlite_amd import gis, synthetic
from atlite_amd.device import Context
CSI = dict(c_temp_amb=1, c_temp_irrad=0.035, r_tmod=298, r_irradiance=1000, k_1=-0.017162, k_2=-0.040289,
           k_3=-0.004681, k_4=0.000148, k_5=0.000169, k_6=0.000005, inverter_efficiency=0.9)
T, Y, X, N = 2000, 200, 200, 100
ctx = Context(0)
inputs, coords = synthetic.pv_inputs(ctx, T, Y, X, offset_hours=4000)
x, y = coords["x"], coords["y"]
dx, dy = x[1]-x[0], y[1]-y[0]
M = gis.compute_indicatormatrix(x, y, gis.random_tessellation(N, (x[0]-dx/2, y[0]-dy/2, x[-1]+dx/2, y[-1]+dy/2)))
plan = ctx.plan(M, row_len=X); S = Y*X
scal = dict(CSI, slope=np.radians(30.0), azimuth=np.radians(180.0))
def timed(inp, reps=7):
    ctx.set_profiling(True); ms=[]
    for i in range(reps+2):
        out = ctx.pv(inp, scal, T, S, plan=plan, options=dict(night_skip=False)); t = ctx.last_kernel_ms()
        if i>=2: ms.append(t)
    return float(np.median(ms)), out
ms0, o0 = timed(inputs); print("clean          %.3f ms" % ms0)
h = inputs["influx_direct"].numpy()
rng = np.random.default_rng(0)
for frac in (1e-5, 1e-3, 0.05):
    hh = h.copy(); m = rng.random(hh.shape) < frac; hh[m] = np.nan
    inp = dict(inputs); inp["influx_direct"] = ctx.upload(hh)
    ms, o = timed(inp); print("NaN frac %-7g %.3f ms  (x%.2f)  NaN outputs: %d" % (frac, ms, ms/ms0, int(np.isnan(o.numpy()).sum())))
