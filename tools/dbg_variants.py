import sys; sys.path.insert(0,'.')
import numpy as np
from digital_earth_amd.renderer import Renderer
r = Renderer((32,16),(0,1,0),texture_source="constant"); r.set_fixed_wavelength(550.0); r.set_topo_res_override(21600)
out={}
for v in (1,2):
    r.set_kernel_variant(v); out[v]=r.debug_samples(0)
d = (out[1].view(np.uint32)!=out[2].view(np.uint32)).any(axis=2)
print('mismatching pixels', d.sum(), 'of', d.size)
ys,xs = np.nonzero(d)
for y,x in list(zip(ys,xs))[:6]: print((x,y), out[1][y,x], out[2][y,x])
r2 = Renderer((128,64),(0,1,0),texture_source="synthetic", texture_size=(1024,512)); r2.set_fov(0.4)
for v in (1,2):
    r2.set_kernel_variant(v); out[v]=r2.debug_samples(0)
d = (out[1].view(np.uint32)!=out[2].view(np.uint32)).any(axis=2)
print('textured mismatching pixels', d.sum(), 'of', d.size)
ys,xs = np.nonzero(d)
for y,x in list(zip(ys,xs))[:6]: print((x,y), out[1][y,x], out[2][y,x])
