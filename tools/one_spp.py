import sys, time; sys.path.insert(0,'.')
from digital_earth_amd.renderer import Renderer
r = Renderer((1920,1080),(0,1,0),texture_source="synthetic"); r.copy_textures()
for spp in (1, 4, 16):
    r.reset_framebuffer(); r.accumulate(spp); r.synchronize()
    t0=time.perf_counter(); n=64//spp
    for i in range(n): r.accumulate(spp)
    r.synchronize(); t1=time.perf_counter()
    print("spp per launch %2d: %d launches, %.1f ms total, %.2f ms per launch, %.0f Msamples/s" % (spp, n, (t1-t0)*1e3, (t1-t0)*1e3/n, 1920*1080*64/(t1-t0)/1e6))
t0=time.perf_counter(); r.reset_framebuffer(); r.accumulate(1); img=r.fetch_image(); t1=time.perf_counter()
print("reset + accumulate(1) + fetch_image (host copy): %.2f ms" % ((t1-t0)*1e3))
