#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

PROVENANCE: these vectors are produced BY THIS REPO'S CPU ORACLE (oracle/), not by the reference: the reference
(Taichi + Vulkan + unshipped NASA maps) cannot run in the build container and holds no golden vectors of its own
(SURVEY §8c).  They pin the oracle against regressions and give the GPU tests a fixture that does not depend on the
oracle library being present.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import make_oracle, load_preset   # noqa: E402
from oracle import oracle_binding as ob        # noqa: E402
from digital_earth_amd import luts             # noqa: E402

SYNTH = (256, 128, 20240127)


def lut_arrays():
    names, crf = luts.load_crfs()
    return luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf, names


def synth_texels():
    o = ob.Oracle(16, 8)
    out = []
    for s in range(7):
        o.generate_texture(s, SYNTH[0], SYNTH[1], SYNTH[2], 0)
        out.append(o.download_texture(s, SYNTH[0], SYNTH[1], 3 if s in (0, 6) else 1))
    return out


def main():
    L = lut_arrays()
    # 1. BASELINE configs[0] in miniature: flat Earth, 550 nm, 32x16, 2 spp
    o = make_oracle(32, 16, L)
    p = o.get_params(); p.flags = 1; p.fixed_wavelength = 550.0; p.topo_res_override = 21600; o.set_params(p)
    o.accumulate(2, 0)
    contract = np.int32(ob.lib().deo_arithmetic_contract())      # the vectors below belong to this arithmetic contract
    np.savez_compressed(os.path.join(HERE, "cfg1_32x16_2spp.npz"), hdr=o.fetch_hdr(), image=o.fetch_image(), trace0=o.debug_samples(0, 0), contract=contract)
    # 2. spectral + synthetic maps, default camera widened so the limb is in frame, and the florida preset
    tex = synth_texels()
    o = make_oracle(64, 32, L, tex)
    p = o.get_params(); p.fov = 0.45; o.set_params(p)
    o.accumulate(2, 7)
    np.savez_compressed(os.path.join(HERE, "spectral_64x32_2spp.npz"), hdr=o.fetch_hdr(), image=o.fetch_image())
    pre = load_preset("config - florida.txt")
    o = make_oracle(64, 32, L, tex)
    p = o.get_params()
    for i in range(3):
        p.camera_pos[i] = pre["pos"][i]; p.look_at[i] = pre["look_at"][i]
    up = np.array(pre["up"], np.float32); up = (np.float32(1.0) / np.sqrt((up[0] * up[0] + up[1] * up[1]) + up[2] * up[2])) * up   # ti.Vector.normalized(): invlen * v
    for i in range(3):
        p.up[i] = float(up[i])
    p.fov = pre["fov"]; p.aspect_scale = pre["aspect_scale"]; p.exposure = pre["exposure"]; p.selected_crf = pre["crf"]
    p.gamma = pre["gamma"]; p.sun_angle = pre["sun_angle"]; p.sun_path_rot = pre["sun_path_rot"]
    o.set_params(p)
    o.accumulate(2, 7)
    np.savez_compressed(os.path.join(HERE, "florida_64x32_2spp.npz"), hdr=o.fetch_hdr(), image=o.fetch_image())
    # 3. leaf math on fixed inputs
    rng = np.random.default_rng(42)
    x = rng.uniform(-20, 20, 2000).astype(np.float32)
    xp = np.exp(rng.uniform(-30, 30, 2000)).astype(np.float32)
    yx = rng.normal(size=(2000, 2)).astype(np.float32)
    xa = rng.uniform(-1, 1, 2000).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "math.npz"), x=x, xp=xp, yx=yx, xa=xa, exp=ob.evaluate("exp", x)[:, 0], log=ob.evaluate("log", xp)[:, 0],
                        sin=ob.evaluate("sin", x)[:, 0], cos=ob.evaluate("cos", x)[:, 0], atan2=ob.evaluate("atan2", yx)[:, 0],
                        asin=ob.evaluate("asin", xa)[:, 0])
    # 4. a slice of every synthetic map (row 37 and 100)
    np.savez_compressed(os.path.join(HERE, "synth_rows.npz"), **{"slot%d" % s: tex[s][[37, 100]] for s in range(7)})
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
