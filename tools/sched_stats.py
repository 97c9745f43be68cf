import sys; sys.path.insert(0,'.')
from digital_earth_amd.renderer import Renderer
r = Renderer((1920,1080),(0,1,0),texture_source="synthetic")
r.copy_textures(); r.enable_counters(True); import os; r.accumulate(int(os.environ.get("SPP","8"))); r.synchronize()
c = r.counters(); s = c["sched"]; n=c["samples"]
print(c)
trips, passes, tapl, gasl, pendl, tapt, gast = s[:7]
print("per-wave-trip: tap lanes %.1f gas lanes %.1f ; passes/trip %.3f ; pend lanes per pass %.1f; tap trips frac %.2f gas trips frac %.2f" % (tapl/trips, gasl/trips, passes/trips, pendl/max(passes,1), tapt/trips, gast/trips))
print("done lanes/trip %.1f surf-wait lanes/trip %.1f" % (s[7]/trips, s[8]/trips))
print("trips per sample-lane: %.1f (ideal %.1f)" % (trips*64/n, (c["sphere_steps"]+c["tracking_steps"])/n))
