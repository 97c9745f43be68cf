"""Headless `EarthViewer` / `Camera` — earth_viewer.py:23-318 without the GGUI window.

The reference's viewer is an interactive ti.ui window (WASD/mouse camera, sliders, 'i'/'o' config.txt save/load,
'p' screenshot).  A headless MI355X node has no window, so the class keeps the reference's names, camera math and
config-file format, and replaces the event loop by `render(spp)`:

    v = EarthViewer(config="config - florida.txt")   # or EarthViewer() for the default camera
    img = v.render(spp=256)                           # accumulate() x spp -> fetch_image()
    v.save("florida.png")
"""
import math
import os

import numpy as np

from . import luts
from .renderer import Renderer

SCREEN_RES = (1920, 1080)      # earth_viewer.py:12
TARGET_FPS = 30                # earth_viewer.py:13
UP_DIR = (0, 1, 0)             # earth_viewer.py:14
planet_r = 6371e3              # lib/volume_rendering_models.py:34


def np_normalize(v):           # lib/math_utils.py:78-80
    return v / np.sqrt(np.sum(v ** 2))


def np_rotate_matrix(axis, theta):
    """lib/math_utils.py:83-102 — rotation matrix for a counter-clockwise rotation about `axis` by `theta` radians
    (Euler-Rodrigues formula).  The reference credits https://stackoverflow.com/a/6802723 for this helper; it is kept
    because SURVEY §8(f)-1 asks for the reference's Camera math."""
    axis = np_normalize(axis)
    a = math.cos(theta / 2.0)
    b, c, d = -axis * math.sin(theta / 2.0)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    bc, ad, ac, ab, bd, cd = b * c, a * d, a * c, a * b, b * d, c * d
    return np.array([[aa + bb - cc - dd, 2 * (bc + ad), 2 * (bd - ac), 0],
                     [2 * (bc - ad), aa + cc - bb - dd, 2 * (cd + ab), 0],
                     [2 * (bd + ac), 2 * (cd - ab), aa + dd - bb - cc, 0],
                     [0, 0, 0, 1]])


class Config:
    """The 10-line preset the reference writes with 'i' and reads with 'o' (earth_viewer.py:100-126, 203-236):
    position, look_at, up (3 floats each), fov, aspect_scale, exposure, selected_crf, gamma, sun_angle, sun_path_rot."""

    def __init__(self, position, look_at, up, fov, aspect_scale, exposure, selected_crf, gamma, sun_angle, sun_path_rot):
        self.position = np.array(position, dtype=np.float64)
        self.look_at = np.array(look_at, dtype=np.float64)
        self.up = np.array(up, dtype=np.float64)
        self.fov, self.aspect_scale, self.exposure = float(fov), float(aspect_scale), float(exposure)
        self.selected_crf, self.gamma = int(selected_crf), float(gamma)
        self.sun_angle, self.sun_path_rot = float(sun_angle), float(sun_path_rot)

    def apply(self, renderer):
        renderer.set_camera_pos(*self.position)
        renderer.set_look_at(*self.look_at)
        renderer.set_up(*self.up)
        renderer.set_fov(self.fov)
        renderer.set_aspect_scale(self.aspect_scale)
        renderer.set_exposure(self.exposure)
        renderer.set_crf(self.selected_crf)
        renderer.set_gamma(self.gamma)
        renderer.set_sun_angle(self.sun_angle)
        renderer.set_sun_path_rot(self.sun_path_rot)

    def write(self, path):
        with open(path, "w") as f:
            for v in (self.position, self.look_at, self.up):
                f.write("%s %s %s\n" % (str(v[0]), str(v[1]), str(v[2])))
            f.write("\n".join(str(x) for x in (self.fov, self.aspect_scale, self.exposure, self.selected_crf, self.gamma,
                                               self.sun_angle)) + "\n")
            f.write(str(self.sun_path_rot))


def load_config(path):
    if not os.path.exists(path):
        alt = os.path.join(luts.CONFIG_DIR, path)
        if os.path.exists(alt):
            path = alt
    with open(path) as f:
        pos = [float(x) for x in f.readline().split()]
        look = [float(x) for x in f.readline().split()]
        up = [float(x) for x in f.readline().split()]
        fov = float(f.readline())
        aspect_scale = float(f.readline())
        exposure = float(f.readline())
        crf = int(f.readline())
        gamma = float(f.readline())
        sun_angle = float(f.readline())
        sun_path_rot = float(f.readline())
    return Config(pos, look, up, fov, aspect_scale, exposure, crf, gamma, sun_angle, sun_path_rot)


class Camera:
    """earth_viewer.py:23-163.  The window argument is kept for signature parity and may be None; movement that the
    reference reads from keys / the mouse is exposed as methods."""

    def __init__(self, window=None, up=UP_DIR):
        self._window = window
        self._lookat_pos = np.array((0.0, 0.0, 0.0))
        self._camera_pos = np.array((-15000000., 0.0, 15000000.))          # :27
        self._up = np_normalize(np.array(up, dtype=np.float64))
        self._last_mouse_pos = None
        self._moved = False

    @property
    def mouse_exclusive_owner(self):
        return True

    def set_up(self, new_up):
        self._up = new_up

    @property
    def position(self):
        return self._camera_pos

    @property
    def look_at(self):
        return self._lookat_pos

    @property
    def target_dir(self):
        return np_normalize(self.look_at - self.position)

    def _compute_cam_r(self):
        return np.sqrt(np.sum(self._camera_pos ** 2))

    def _compute_left_dir(self, tgtdir):                                   # :157-161
        cos = np.dot(self._up, tgtdir)
        if abs(cos) > 0.999:
            return np.array([-1.0, 0.0, 0.0])
        return np.cross(self._up, tgtdir)

    def rotate(self, dx, dy, scale=3):
        """_update_by_mouse (:43-67) for a cursor displacement (dx, dy)."""
        out_dir = self._lookat_pos - self._camera_pos
        leftdir = self._compute_left_dir(np_normalize(out_dir))
        rotx = np_rotate_matrix(self._up, dx * scale)
        roty = np_rotate_matrix(leftdir, dy * scale)
        out_dir_homo = np.array(list(out_dir) + [0.0])
        new_out_dir = np.matmul(np.matmul(roty, rotx), out_dir_homo)[:3]
        self._lookat_pos = self._camera_pos + new_out_dir
        self._moved = True
        return True

    def move(self, direction, elapsed_time, shift=False):
        """_update_by_wasd (:72-144) for a summed key direction vector."""
        d = np.array(direction, dtype=np.float64) * 0.05
        speed = 30.0 * max(min(self._compute_cam_r() - planet_r, planet_r * 0.5), 0.0)
        if shift:
            speed *= 3.0
        cam_step = d * speed * elapsed_time
        self._lookat_pos += cam_step
        self._camera_pos += cam_step
        if self._compute_cam_r() < planet_r * 1.000:
            self._lookat_pos -= cam_step * 2
            self._camera_pos -= cam_step * 2
        self._moved = True
        return True

    def update_camera(self, elapsed_time):
        """earth_viewer.py:146-155: True when the camera changed since the last call.  No input devices on a headless node:
        changes come from rotate() / move() / load()."""
        moved, self._moved = self._moved, False
        return moved

    def load(self, cfg):
        self._camera_pos = cfg.position.copy()
        self._lookat_pos = cfg.look_at.copy()
        self._up = cfg.up.copy()
        self._moved = True


class EarthViewer:
    """earth_viewer.py:166-318, headless."""

    def __init__(self, config=None, screen_res=SCREEN_RES, **renderer_kwargs):
        self.window = None
        self.camera = Camera(self.window, up=UP_DIR)
        self.renderer = Renderer(image_res=screen_res, up=UP_DIR, **renderer_kwargs)
        self.renderer.set_camera_pos(*self.camera.position)
        self.renderer.copy_textures()
        self.config = None
        if config is not None:
            self.load_config(config)
        self._image = None
        self._sliders = None

    def load_config(self, path_or_config):
        cfg = path_or_config if isinstance(path_or_config, Config) else load_config(path_or_config)
        self.config = cfg
        self.camera.load(cfg)
        self.camera.update_camera(0.0)          # the renderer gets the camera right here
        cfg.apply(self.renderer)
        self._sliders = None                    # 'o' re-reads the slider values from the file (:213-236)
        self.renderer.reset_framebuffer()

    def save_config(self, path):
        r = self.renderer
        Config(self.camera.position, self.camera.look_at, self.camera._up, r.fov[None], r.aspect_scale[None], r.exposure[None],
               r.selected_crf[None], r.gamma[None], r.sun_angle[None], r.sun_path_rot[None]).write(path)

    def render(self, spp=1, chunk=64):
        """The body of the reference loop (:241-243): accumulate() spp times, then fetch_image()."""
        left = int(spp)
        while left > 0:
            n = min(left, chunk)
            self.renderer.accumulate(n)
            left -= n
        self._image = self.renderer.fetch_image()
        return self._image

    def frame(self, spp=1, copy=True, pipelined=False, **sliders):
        """ONE iteration of the reference's window loop (earth_viewer.py:203-317), with the GUI sliders passed as keywords
        (sun_angle, sun_path_rot, fov, aspect_scale, exposure, selected_crf, gamma):
          1. a moved camera is pushed to the renderer and marks the framebuffer for reset          (:206-213)
          2. accumulate() x spp, fetch_image() — with the parameters of the PREVIOUS iteration     (:241-243)
          3. slider changes: sun angle / sun path rotation / fov / aspect scale mark the framebuffer for reset;
             exposure, camera response and gamma do not (display-only)                            (:268-305)
          4. the scalars are written to the renderer, then the reset happens                       (:308-317)
        Returns the displayed image of step 2 (copy=False: a read-only view that the next frame() overwrites — what a canvas needs).
        pipelined=True (or 2: two frames of lag): the display and the host copy of this frame are only enqueued (Renderer.fetch_image(lag=...)) and the
        image RETURNED is the previous iteration's (None in the first): the next iteration's accumulate() renders while this frame is shown — the same images, one
        iteration later, at a fraction of the frame time; finish() returns the last one."""
        r = self.renderer
        should_reset = False
        if self.camera.update_camera(0.0):
            r.set_camera_pos(*self.camera.position)
            r.set_look_at(*self.camera.look_at)
            r.set_up(*self.camera._up)
            should_reset = True
        r.accumulate(int(spp))                  # == accumulate() x spp, bit for bit
        self._image = r.fetch_image(copy=copy, lag=int(pipelined))
        if self._sliders is None:               # current_* of earth_viewer.py:191-199, read once when the loop starts
            self._sliders = {"sun_angle": r.sun_angle[None], "sun_path_rot": r.sun_path_rot[None], "fov": r.fov[None],
                             "aspect_scale": r.aspect_scale[None], "exposure": r.exposure[None],
                             "selected_crf": r.selected_crf[None], "gamma": r.gamma[None]}
        cur = self._sliders
        for k, v in sliders.items():
            if k not in cur:
                raise TypeError("unknown slider %r" % k)
            if v != cur[k]:
                if k in ("sun_angle", "sun_path_rot", "fov", "aspect_scale"):
                    should_reset = True
                cur[k] = v
        r.sun_angle[None] = cur["sun_angle"]; r.sun_path_rot[None] = cur["sun_path_rot"]
        r.fov[None] = cur["fov"]; r.aspect_scale[None] = cur["aspect_scale"]
        r.exposure[None] = cur["exposure"]; r.gamma[None] = cur["gamma"]; r.selected_crf[None] = cur["selected_crf"]
        if should_reset:
            r.reset_framebuffer()
        return self._image

    def finish(self, copy=True):
        """End a pipelined loop: the image of the last frame() iteration (None when nothing is in flight)."""
        img = self.renderer.fetch_pending(copy=copy)
        if img is not None:
            self._image = img
        return img

    def close(self):
        """Release the renderer.  A zero-copy image kept from frame(copy=False) is copied first: it lives in the renderer's staging buffer."""
        if self._image is not None and getattr(self._image, "_owner", None) is not None:
            self._image = np.array(self._image)
        self.renderer.close()

    def save(self, path):
        """'p' screenshot (:244-250): writes the displayed image; `.npy` keeps the float (W, H, 3) array."""
        if self._image is None:
            self.render(1)
        if path.endswith(".npy"):
            np.save(path, self._image)
            return
        from PIL import Image
        a = (np.clip(self._image, 0.0, 1.0) * 255).astype(np.uint8)        # Renderer.to_vec3u
        Image.fromarray(a.transpose(1, 0, 2)[::-1]).save(path)

    def start(self, spp=64, out="screenshot/earth.png"):
        """Reference entry point (main.py:4).  Headless: render one frame and save it."""
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        self.render(spp)
        self.save(out)
        return out
