"""Texture set of the reference — lib/textures.py:1-79 (file names and resolutions by TEXTURE_QUALITY) — plus the
synthetic stand-ins used when the NASA maps are absent (the reference does not ship them: README.md:31-32).

Slot numbering matches include/digital_earth.h (DE_TEX_*)."""
import os
import numpy as np

TEX_RES_4K = (3840, 1920)      # lib/textures.py:2-6
TEX_RES_8K = (8100, 4050)
TEX_RES_10K = (10800, 5400)
TEX_RES_16K = (16200, 8100)
TEX_RES_21K = (21600, 10800)

ALBEDO, TOPOGRAPHY, OCEAN, CLOUDS, BATHYMETRY, EMISSIVE, STARS = range(7)
SLOT_NAMES = ("albedo", "topography", "ocean", "clouds", "bathymetry", "emissive", "stars")
SLOT_CHANNELS = (3, 1, 1, 1, 1, 1, 3)

# (file, resolution) per slot and quality tier — lib/textures.py:34-79.
# Quality 0 in the reference never defines CLOUDS_TEX_RES (NameError at renderer.py:76); 4K is used here.
_Q = {
    0: {ALBEDO: ("earth_color_4K.png", TEX_RES_4K), TOPOGRAPHY: ("topography_4K.png", TEX_RES_4K),
        OCEAN: ("earth_landocean_4K.png", TEX_RES_4K), CLOUDS: ("earth_clouds_4K.png", TEX_RES_4K),
        BATHYMETRY: ("earth_bathymetry_4k.png", TEX_RES_4K), EMISSIVE: ("earth_nightlights_4K.png", TEX_RES_4K),
        STARS: ("stars_8K.jpg", TEX_RES_8K)},
    1: {ALBEDO: ("earth_color_10K.png", TEX_RES_10K), TOPOGRAPHY: ("topography_10K.png", TEX_RES_10K),
        OCEAN: ("earth_landocean_8K.png", TEX_RES_8K), CLOUDS: ("earth_clouds_8K.png", TEX_RES_8K),
        BATHYMETRY: ("earth_bathymetry_10k.png", TEX_RES_10K), EMISSIVE: ("earth_nightlights_10K.png", TEX_RES_10K),
        STARS: ("stars_16K.png", TEX_RES_16K)},
    2: {ALBEDO: ("earth_color_21K.png", TEX_RES_21K), TOPOGRAPHY: ("topography_21K.png", TEX_RES_21K),
        OCEAN: ("earth_landocean_16K.png", TEX_RES_16K), CLOUDS: ("earth_clouds_21K.png", TEX_RES_21K),
        BATHYMETRY: ("earth_bathymetry_21k.png", TEX_RES_21K), EMISSIVE: ("earth_nightlights_21K.png", TEX_RES_21K),
        STARS: ("stars_16K.png", TEX_RES_16K)},
}
TEXTURE_QUALITY = 2            # lib/textures.py:1
SYNTH_SEED = 20240127


def texture_table(quality=TEXTURE_QUALITY):
    """{slot: (file name, (width, height))} for a quality tier."""
    return dict(_Q[quality])


def load_image_texels(path, channels):
    """ti.tools.imread equivalent for one map: returns uint8 [height][width][channels] with row 0 = v 0 = south
    (Taichi's imread is [x][y] with y up, renderer.py:63-93; only channel 0 of the grey maps is used :68-88)."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    img = Image.open(path)
    img = img.convert("RGB")
    a = np.asarray(img)[::-1, :, :]           # image rows are top-down; flip so that row 0 is the south edge
    if channels == 1:
        a = a[:, :, 0:1]
    return np.ascontiguousarray(a)


def constant_texels(slot, value):
    """1x1 map — BASELINE cfg1's "flat-albedo procedural Earth (no textures)"."""
    ch = SLOT_CHANNELS[slot]
    v = np.broadcast_to(np.asarray(value, dtype=np.uint8), (ch,))
    return np.ascontiguousarray(v.reshape(1, 1, ch))


def find_texture_dir():
    """textures/ next to the working directory, as the reference expects (lib/textures.py:10-27)."""
    for d in (os.path.join(os.getcwd(), "textures"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "textures")):
        if os.path.isdir(d):
            return d
    return None
