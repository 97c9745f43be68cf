import sys; sys.path.insert(0,'.')
from digital_earth_amd.renderer import Renderer
r = Renderer((1920,1080),(0,1,0),texture_source="synthetic")
r.copy_textures(); r.enable_counters(True); import os; r.accumulate(int(os.environ.get("SPP","8"))); r.synchronize()
c = r.counters(); s = c["sched"]; n=c["samples"]
print(c)
trips, passes = s[0], s[1]
tm = s[2:9]; tot = sum(tm)
names = ["loop head+trigger", "TAP body", "GAS body", "cheap transitions", "END", "stash+heavy batches", "NEW+VERTEX"]
print("trips per sample-lane %.1f (ideal %.1f), passes/trip %.3f" % (trips*64/n, (c["sphere_steps"]+c["tracking_steps"])/n, passes/trips))
for nm, x in zip(names, tm): print("  %-22s %5.1f %%  (%.0f cycles per trip)" % (nm, 100.0*x/tot, x/trips))
