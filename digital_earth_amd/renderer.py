"""`Renderer` — the host-side operator API of the reference (renderer.py:15-401), bound to the gfx950 library.

Method and attribute names, argument meaning and call order are the reference's:

    r = Renderer(image_res=(1920, 1080), up=(0, 1, 0))      # renderer.py:17
    r.set_camera_pos(x, y, z); r.set_look_at(...)            # :225-266
    r.copy_textures()                                        # :136
    r.accumulate()                                           # :371   (1 sample per pixel, current_spp += 1)
    img = r.fetch_image()                                    # :382   (W, H, 3) float32 in [0, 1]
    r.fov[None] = 0.2                                        # scalar fields keep Taichi's [None] indexing

All device work happens in libdigitalearth_hip.so through ctypes (digital_earth_amd/_native.py).  There is no
Taichi and no CPU implementation in this package.
"""
import ctypes
import weakref
import os

import numpy as np

from . import _native, luts, textures as tex
from ._native import DeParams, DeCounters, check


class _ScalarField:
    """Stand-in for a 0-d ti.field: `field[None]` reads, `field[None] = v` writes (earth_viewer.py:191-199,308-314)."""

    def __init__(self, owner, name, cast):
        # a weak reference: the Renderer must be freed (and with it its device memory: up to tens of GB of pipeline queues) the
        # moment the last user reference goes, not at the garbage collector's next cycle sweep
        self._owner, self._name, self._cast = weakref.proxy(owner), name, cast

    def __getitem__(self, key):
        return self._cast(getattr(self._owner._params, self._name))

    def __setitem__(self, key, value):
        setattr(self._owner._params, self._name, self._cast(value))
        self._owner._push_params()


class _VectorField:
    def __init__(self, owner, name):
        self._owner, self._name = weakref.proxy(owner), name

    def __getitem__(self, key):
        return np.array(list(getattr(self._owner._params, self._name)), dtype=np.float32)

    def __setitem__(self, key, value):
        arr = getattr(self._owner._params, self._name)
        for i in range(3):
            arr[i] = float(value[i])
        self._owner._push_params()


class _StagingView(np.ndarray):
    """fetch_image(copy=False): a read-only window on the library's pinned staging buffer.  Overwritten by the next fetch_* call of its
    Renderer; holds a reference to that Renderer so that garbage collection cannot free the buffer under it."""
    _owner = None

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


class Renderer:
    """GPU implementation of the reference `Renderer` (renderer.py:15-401).

    Extra keyword arguments (no reference counterpart):
      device          HIP device index (one context = one GPU)
      texture_source  "auto": image files from `texture_dir` when present, else synthetic stand-ins;
                      "files": image files only (FileNotFoundError like the reference when absent);
                      "synthetic": procedural maps generated on the GPU (the reference ships no maps, README.md:31-32);
                      "constant": 1x1 maps (BASELINE cfg1: flat albedo, no topography, no clouds)
      texture_quality 0/1/2 — lib/textures.py:1 (resolutions of the tier are used for synthetic maps too)
      texture_size    override (w, h) for every synthetic map, or {slot: (w, h)} per map (tests)
      cloud_heavy     synthetic cloud variant of BASELINE cfg4
      seed            base seed of the per-sample RNG streams
    """

    def __init__(self, image_res, up, device=0, texture_source="auto", texture_dir=None,
                 texture_quality=tex.TEXTURE_QUALITY, texture_size=None, cloud_heavy=False, seed=0,
                 synth_seed=tex.SYNTH_SEED):
        self._lib = _native.load()
        self.image_res = (int(image_res[0]), int(image_res[1]))
        self.aspect_ratio = image_res[0] / image_res[1]                  # renderer.py:19
        self.current_spp = 0                                             # :23
        self.seed = int(seed)
        self.tile_rank, self.tile_world = 0, 1
        self._h = ctypes.c_void_p()
        check(self._lib.de_create(int(device), self.image_res[0], self.image_res[1], ctypes.byref(self._h)))
        _native.apply_env_tuning(self._h)      # experiment overrides (DE_KERNEL, DE_V6_* ...): read HERE, not in the library
        self._params = DeParams()
        check(self._lib.de_get_params(self._h, ctypes.byref(self._params)))   # reference defaults, renderer.py:20-22,49-58

        self.fov = _ScalarField(self, "fov", float)                      # :27
        self.aspect_scale = _ScalarField(self, "aspect_scale", float)    # :28
        self.exposure = _ScalarField(self, "exposure", float)            # :30
        self.selected_crf = _ScalarField(self, "selected_crf", int)      # :31
        self.gamma = _ScalarField(self, "gamma", float)                  # :33
        self.sun_angle = _ScalarField(self, "sun_angle", float)          # :36
        self.sun_path_rot = _ScalarField(self, "sun_path_rot", float)    # :37
        self.camera_pos = _VectorField(self, "camera_pos")               # :39
        self.look_at = _VectorField(self, "look_at")                     # :40
        self.up = _VectorField(self, "up")                               # :41

        self.set_up(*up)                                                 # :49
        self.set_fov(np.radians(27.) * 0.5)                              # :50
        self.set_aspect_scale(1.0)
        self.set_exposure(2.5)
        self.set_gamma(1.0)
        self.set_crf(0)
        self.set_sun_angle(np.radians(60.0))
        self.set_sun_path_rot(np.radians(-45.0))

        # textures (renderer.py:60-94): staged here, sent to the device by copy_textures()
        self._texture_plan = self._plan_textures(texture_source, texture_dir, texture_quality, texture_size,
                                                 cloud_heavy, synth_seed)
        self._textures_copied = False
        self._bound = None           # (tensor, stream) kept alive while the context points at them (parallel.DistributedFrame)

        # LUTs (renderer.py:96-134)
        self.crf_names = []
        crf_array = self.load_crfs()
        self.crf_lut_res = (1024, len(self.crf_names))
        self._luts = (luts.load_cie(), luts.load_srgb2spec(), luts.load_o3(), crf_array)
        self.set_crf_count(self.crf_lut_res[1])

    # ------------------------------------------------------------------ lifetime
    def close(self):
        """de_destroy.  Fails (DigitalEarthError, DE_ERR_STATE) while another Renderer still borrows this one's maps, and (RuntimeError)
        while a zero-copy image of fetch_image(copy=False) is still referenced: that array IS the context's pinned staging buffer."""
        if getattr(self, "_h", None):
            ref = getattr(self, "_view_ref", None)
            if ref is not None and ref() is not None:
                raise RuntimeError("a fetch_image(copy=False) view of this Renderer is still referenced: drop it (or copy it) before close()")
            check(self._lib.de_destroy(self._h))
            self._h = ctypes.c_void_p()
            self._lender = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ attributes kept from the reference
    @property
    def vignette_strength(self):
        return self._params.vignette_strength

    @vignette_strength.setter
    def vignette_strength(self, v):
        self._params.vignette_strength = float(v)
        self._push_params()

    @property
    def vignette_radius(self):
        return self._params.vignette_radius

    @vignette_radius.setter
    def vignette_radius(self, v):
        self._params.vignette_radius = float(v)
        self._push_params()

    @property
    def vignette_center(self):
        return [self._params.vignette_center[0], self._params.vignette_center[1]]

    @vignette_center.setter
    def vignette_center(self, v):
        self._params.vignette_center[0], self._params.vignette_center[1] = float(v[0]), float(v[1])
        self._push_params()

    @property
    def land_height_scale(self):
        return self._params.land_height_scale

    @land_height_scale.setter
    def land_height_scale(self, v):
        self._params.land_height_scale = float(v)
        self._push_params()

    @property
    def color_buffer(self):
        """HDR accumulation as a (W, H, 3) float32 array (renderer.py:25,330)."""
        return self.fetch_hdr()

    def _push_params(self):
        check(self._lib.de_set_params(self._h, ctypes.byref(self._params)))

    # ------------------------------------------------------------------ textures
    def _plan_textures(self, source, texture_dir, quality, size, cloud_heavy, synth_seed):
        table = tex.texture_table(quality)
        if source in ("auto", "files"):
            d = texture_dir or tex.find_texture_dir()
            have = d is not None and all(os.path.exists(os.path.join(d, table[s][0])) for s in table)
            if have:
                return [("file", s, os.path.join(d, table[s][0])) for s in range(7)]
            if source == "files":
                raise FileNotFoundError("texture files %s not found in %r (the reference expects them under textures/, "
                                        "README.md:31-32)" % ([table[s][0] for s in table], d))
            source = "synthetic"
        if source == "synthetic":
            plan = []
            for s in range(7):
                w, h = (size[s] if isinstance(size, dict) else size) if size is not None else table[s][1]
                plan.append(("synthetic", s, (int(w), int(h), int(synth_seed), 1 if cloud_heavy else 0)))
            return plan
        if source == "constant":
            return [("array", s, tex.constant_texels(s, (128, 128, 128) if s == tex.ALBEDO else 0)) for s in range(7)]
        raise ValueError("texture_source must be auto, files, synthetic or constant")

    def set_texture(self, slot, texels):
        """Replace one map by a uint8 array [height][width][channels] (row 0 = south)."""
        self._texture_plan[slot] = ("array", slot, np.ascontiguousarray(texels, dtype=np.uint8))
        if self._textures_copied:
            self._copy_one(self._texture_plan[slot])

    def _copy_one(self, item):
        kind, slot, arg = item
        if kind == "synthetic":
            w, h, seed, variant = arg
            check(self._lib.de_generate_texture(self._h, slot, w, h, seed, variant))
            return
        arr = tex.load_image_texels(arg, tex.SLOT_CHANNELS[slot]) if kind == "file" else arg
        h, w, ch = arr.shape
        check(self._lib.de_upload_texture(self._h, slot, arr.ctypes.data, w, h, ch))

    def copy_textures(self):
        """renderer.py:136-145 — the nine copy_* upload kernels (7 maps + CIE + CRF)."""
        for item in self._texture_plan:
            self._copy_one(item)
        cie, s2s, o3, crf = self._luts
        check(self._lib.de_upload_luts(self._h, cie.ctypes.data, s2s.ctypes.data, o3.ctypes.data, crf.ctypes.data,
                                       crf.shape[1]))
        self._textures_copied = True

    def copy_texture(self, slot):
        """Send ONE map of this renderer's own texture plan to the device.  After share_textures_from() this makes the renderer the owner
        of that map while it keeps borrowing the others (a second view of the same Earth under another cloud cover: BASELINE cfg4's
        cloud-heavy variant differs from the default in the cloud map alone)."""
        self._copy_one(self._texture_plan[slot])
        self._own_slots = getattr(self, "_own_slots", set()) | {int(slot)}

    def share_textures_from(self, other):
        """Use `other`'s device-resident maps and LUTs (same GPU) instead of uploading / generating copies: a second
        frame in flight costs no second 9 GB.  Keeps a reference to `other` so it outlives this renderer."""
        if not other._textures_copied:
            other.copy_textures()
        check(self._lib.de_share_textures(self._h, other._h))
        self._lender = other
        self._own_slots = set()
        self._textures_copied = True

    def trim_textures(self):
        """Free the as-uploaded copies of the maps (2.1 GB at quality 2); the kernels keep their packed copies."""
        if not self._textures_copied:
            self.copy_textures()
        check(self._lib.de_trim_textures(self._h))

    def download_texture(self, slot):
        """The map as uploaded.  A borrower (share_textures_from) holds no as-uploaded copy: the owner's is returned."""
        if getattr(self, "_lender", None) is not None and int(slot) not in getattr(self, "_own_slots", ()):
            return self._lender.download_texture(slot)
        w, h, ch = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self._lib.de_texture_info(self._h, slot, ctypes.byref(w), ctypes.byref(h), ctypes.byref(ch)))
        out = np.zeros((h.value, w.value, ch.value), dtype=np.uint8)
        check(self._lib.de_download_texture(self._h, slot, out.ctypes.data, out.nbytes))
        return out

    def load_crfs(self):
        """renderer.py:147-167 — (1024, n, 3) float32; fills self.crf_names (Neutral.rf first, then sorted: Q10)."""
        names, arr = luts.load_crfs()
        self.crf_names = list(names)
        return arr

    # ------------------------------------------------------------------ setters, renderer.py:224-266
    def set_camera_pos(self, x, y, z):
        self._params.camera_pos[0], self._params.camera_pos[1], self._params.camera_pos[2] = float(x), float(y), float(z)
        self._push_params()

    def set_up(self, x, y, z):
        v = np.array([x, y, z], dtype=np.float32)
        # ti.Vector.normalized() in f32 (taichi/lang/matrix.py): invlen = 1 / norm; invlen * v — checked against the value the
        # reference's set_up kernel stores (tests/golden/ref_frames_*.npz, tests/test_gpu_ref_fixtures.py)
        v = (np.float32(1.0) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])) * v
        self._params.up[0], self._params.up[1], self._params.up[2] = float(v[0]), float(v[1]), float(v[2])
        self._push_params()

    def set_look_at(self, x, y, z):
        self._params.look_at[0], self._params.look_at[1], self._params.look_at[2] = float(x), float(y), float(z)
        self._push_params()

    def set_fov(self, fov):
        self._params.fov = float(fov)
        self._push_params()

    def set_aspect_scale(self, scale):
        self._params.aspect_scale = float(scale)
        self._push_params()

    def set_exposure(self, exposure):
        self._params.exposure = float(exposure)
        self._push_params()

    def set_gamma(self, gam):
        self._params.gamma = float(gam)
        self._push_params()

    def set_crf(self, index):
        self._params.selected_crf = int(index)
        self._push_params()

    def set_crf_count(self, num):
        self._crf_count = int(num)       # the device takes the count from the uploaded CRF array (renderer.py:257)

    def set_sun_angle(self, ang):
        self._params.sun_angle = float(ang)
        self._push_params()

    def set_sun_path_rot(self, ang):
        self._params.sun_path_rot = float(ang)
        self._push_params()

    # extensions of the scalar state (include/digital_earth.h: de_params.flags)
    def set_fixed_wavelength(self, nm):
        """BASELINE cfg1: trace a single wavelength instead of sampling one per path (None to disable)."""
        if nm is None:
            self._params.flags &= ~_native.DE_FLAG_FIXED_WAVELENGTH
        else:
            self._params.flags |= _native.DE_FLAG_FIXED_WAVELENGTH
            self._params.fixed_wavelength = float(nm)
        self._push_params()

    def set_flag(self, flag, on):
        if on:
            self._params.flags |= flag
        else:
            self._params.flags &= ~flag
        self._push_params()

    def set_integrator(self, name):
        """'path_tracer' (pathtracer.py:316, what the reference runs) or 'ray_marcher' (pathtracer.py:544)."""
        if name not in ("path_tracer", "ray_marcher"):
            raise ValueError(name)
        self.set_flag(_native.DE_FLAG_RAY_MARCHER, name == "ray_marcher")

    def set_display_transform(self, name):
        """'opendrt' (what the reference runs: OpenDRT + camera response + gamma, renderer.py:357-362) or 'agx'
        (the alternative kept behind a comment at renderer.py:356, lib/AgX.py)."""
        if name not in ("opendrt", "agx"):
            raise ValueError(name)
        self.set_flag(_native.DE_FLAG_AGX, name == "agx")

    def set_fast_math(self, on=True):
        """OPT-IN (DE_FLAG_FAST_MATH): path_tracer on the hardware's transcendental units instead of the arithmetic contract's fixed sequences.
        Faster (profiles/r5_fast_math.md) and no longer bit-reproducible against the CPU oracle: a sample whose collision test lands on the other
        side follows a different path, so images agree with the contract's statistically, not sample by sample."""
        self.set_flag(_native.DE_FLAG_FAST_MATH, bool(on))

    def set_topo_res_override(self, res):
        self._params.topo_res_override = int(res)
        self._push_params()

    def set_tile_partition(self, rank, world):
        """Multi-GPU: this context renders only the 8x8 tiles (tx, ty) with (tx + ty) % world == rank."""
        if not (0 <= rank < world):
            raise ValueError("rank %d not in [0, %d)" % (rank, world))
        self.tile_rank, self.tile_world = int(rank), int(world)

    def set_sample_partition(self, rank, world):
        """Multi-GPU, the other split: this context renders, of the frame's sample indices, those = rank (mod world) of every
        pixel it owns; accumulate(spp) still advances the frame's sample counter by spp.  The ranks' buffers are then partial sums of every
        pixel: assemble them with reduce_ordered / parallel.reduce_hdr_ordered (rank-ordered additions), not with a plain sum-reduce."""
        if not (0 <= rank < world):
            raise ValueError("rank %d not in [0, %d)" % (rank, world))
        check(self._lib.de_set_sample_partition(self._h, int(rank), int(world)))
        self.sample_rank, self.sample_world = int(rank), int(world)

    # ------------------------------------------------------------------ frame loop, renderer.py:367-384
    def reset_framebuffer(self):
        self.current_spp = 0
        check(self._lib.de_reset(self._h))

    def accumulate(self, spp=1):
        """renderer.py:371-380, `spp` times in one launch (the reference's accumulate() is spp = 1)."""
        if not self._textures_copied:
            self.copy_textures()
        check(self._lib.de_accumulate(self._h, int(spp), self.seed, self.tile_rank, self.tile_world))
        self.current_spp += int(spp)

    def _staging_view(self, ptr):
        view = np.ctypeslib.as_array(ptr, shape=(self.image_res[0], self.image_res[1], 3)).view(_StagingView)
        view.flags.writeable = False
        view._owner = self                      # the memory belongs to the context: the view keeps its Renderer alive ...
        self._view_ref = weakref.ref(view)      # ... and close() refuses while the view is
        return view

    def fetch_image(self, copy=True, lag=0):
        """renderer.py:382-384 — display transform of the accumulation; (W, H, 3) float32 in [0, 1].  copy=False returns a read-only
        view of the library's pinned staging buffer, valid until the next fetch_* call on this renderer (what a window loop needs: it
        hands the image to the canvas before it renders again) — no 24.9 MB host copy, no fresh pages.

        lag=1, 2 or 3: the window loop PIPELINED (de_fetch_image_begin / _end).  The display transform and the device-to-host copy of the frame as it
        stands are only ENQUEUED; the call returns the image of the lag-th PREVIOUS call (None until there is one), so the caller's next
        accumulate() renders while this frame is displayed and copied.  A lone one-sample launch takes ~10 ms (its longest path), so lag=1 runs
        at ~5.1 ms per 1080p frame, lag=2 at ~4.1 ms, lag=3 at ~3.5 ms (the floor, launches back to back without any fetch, is 2.9 ms).  Every image equals what the synchronous loop returns for the same frame, bit for bit,
        `lag` calls later; fetch_pending() hands out the rest.  A view (copy=False) stays valid until the next fetch_image call."""
        if not self._textures_copied:
            self.copy_textures()
        if lag not in (0, 1, 2, 3):
            raise ValueError("lag must be 0 ... 3")
        if lag:
            check(self._lib.de_fetch_image_begin(self._h))
            self._fetches = getattr(self, "_fetches", 0) + 1
            if self._fetches <= lag:
                return None
            return self._fetch_end(copy)
        if getattr(self, "_fetches", 0):
            raise RuntimeError("lagged fetches are in flight: fetch_pending() first")
        if not copy:
            ptr = ctypes.POINTER(ctypes.c_float)()
            check(self._lib.de_fetch_image_view(self._h, ctypes.byref(ptr)))
            return self._staging_view(ptr)
        out = np.empty((self.image_res[0], self.image_res[1], 3), dtype=np.float32)
        check(self._lib.de_fetch_image(self._h, out.ctypes.data))
        return out

    def _fetch_end(self, copy):
        ptr = ctypes.POINTER(ctypes.c_float)()
        self._fetches -= 1
        check(self._lib.de_fetch_image_end(self._h, ctypes.byref(ptr)))
        view = self._staging_view(ptr)
        return np.array(view) if copy else view

    def fetch_pending(self, copy=True, all_images=False):
        """End the pipelined window loop: wait for the fetches still in flight and return the newest image (None when there is none);
        all_images=True: the list of all of them, oldest first (copies)."""
        imgs = []
        while getattr(self, "_fetches", 0) > 0:
            imgs.append(self._fetch_end(copy or all_images))
        if all_images:
            return imgs
        return imgs[-1] if imgs else None

    def fetch_hdr(self):
        out = np.empty((self.image_res[0], self.image_res[1], 3), dtype=np.float32)
        check(self._lib.de_fetch_hdr(self._h, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ measurement / plumbing
    def synchronize(self):
        check(self._lib.de_synchronize(self._h))

    def last_accumulate_ms(self):
        ms = ctypes.c_float()
        check(self._lib.de_last_accumulate_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def enable_counters(self, on=True):
        check(self._lib.de_enable_counters(self._h, 1 if on else 0))

    def set_kernel_variant(self, variant):
        """4 = automatic (default): a call of at least 4096 paths runs the per-CU stage scheduler (render_kernel_v6: one persistent launch, + a small
        tail launch for big calls), smaller calls and counting launches the wave-level state machine; 2 / 6 = that kernel for every call.  1, 3, 5
        (per-lane loops, wavefront pipeline, HBM-queue scheduler) exist in the legacy library only.  Identical results, bit for bit."""
        check(self._lib.de_set_kernel_variant(self._h, int(variant)))

    def set_memory_budget(self, n_bytes):
        """Bytes the context may hold in pipeline queues / cold records / chunk pools (0 = automatic); include/digital_earth.h."""
        check(self._lib.de_set_memory_budget(self._h, ctypes.c_uint64(int(n_bytes))))

    def memory_use(self):
        n = ctypes.c_uint64()
        check(self._lib.de_get_memory_use(self._h, ctypes.byref(n)))
        return int(n.value)

    def last_call_info(self):
        """What the last accumulate() ran: dict(variant, pipes, depths, launches, kernel) — include/digital_earth.h."""
        v = [ctypes.c_int() for _ in range(4)]
        check(self._lib.de_last_call_info(self._h, *[ctypes.byref(x) for x in v]))
        variant, pipes, depths, launches = (int(x.value) for x in v)
        name = {0: "ray marcher", 1: "per-lane loops", 2: "state machine", 5: "persistent stage scheduler", 6: "per-CU stage scheduler",
                3: "pipeline x%d, %d rounds%s" % (pipes, depths, " + state machine" if depths < 25 else "")}.get(variant, "?")
        return dict(variant=variant, pipes=pipes, depths=depths, launches=launches, kernel=name)

    def last_launch_phases(self):
        """(launch_ms, drain_ms) of the last render_kernel_v6 launch, from the kernel's own clock: the drain is the time between the
        first wave that found no work item left and the last wave's exit (include/digital_earth.h)."""
        ms = (ctypes.c_float * 2)()
        check(self._lib.de_last_launch_phases(self._h, ms))
        return float(ms[0]), float(ms[1])

    def tuning(self):
        """The context's de_tuning (include/digital_earth.h) as a ctypes struct; change fields and pass it to set_tuning."""
        t = _native.DeTuning()
        check(self._lib.de_get_tuning(self._h, ctypes.byref(t)))
        return t

    def set_tuning(self, t):
        check(self._lib.de_set_tuning(self._h, ctypes.byref(t)))

    def v5_stats(self, n=24):
        """Statistics block of the persistent stage scheduler (legacy library, kernel variant 5 with DE_V5_STATS=1); include/digital_earth_legacy.h."""
        if not hasattr(self._lib, "de_debug_v5_stats"):
            raise RuntimeError("de_debug_v5_stats lives in the legacy library (DE_LIB_PATH=.../libdigitalearth_hip_legacy.so)")
        out = (ctypes.c_uint64 * n)()
        check(self._lib.de_debug_v5_stats(self._h, out, n))
        return [int(x) for x in out]

    def v6_stats(self, n=48):
        """Statistics block of the per-CU stage scheduler (kernel variant 6 with DE_V6_STATS=1); include/digital_earth.h."""
        out = (ctypes.c_uint64 * n)()
        check(self._lib.de_debug_v6_stats(self._h, out, n))
        return [int(x) for x in out]

    def counters(self):
        c = DeCounters()
        check(self._lib.de_get_counters(self._h, ctypes.byref(c)))
        return c.as_dict()

    def sched_stats(self, n=48):
        """Raw scheduler statistics of render_kernel_v2 (counters must be enabled); see tools/sched_stats.py."""
        out = (ctypes.c_uint64 * n)()
        check(self._lib.de_debug_sched_stats(self._h, out, n))
        return [int(x) for x in out]

    def hdr_device_pointer(self):
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        check(self._lib.de_hdr_device_ptr(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def bind_hdr(self, device_ptr, n_floats):
        """Accumulate into caller-owned device memory (e.g. a torch tensor) instead of the library's buffer."""
        check(self._lib.de_bind_hdr(self._h, ctypes.c_void_p(device_ptr), int(n_floats)))

    def unbind(self):
        """Back to the library's own HDR buffer and stream (undoes bind_hdr / set_stream)."""
        check(self._lib.de_bind_hdr(self._h, None, 0))
        check(self._lib.de_use_own_stream(self._h))
        self._bound = None

    # in-library collective (include/digital_earth.h: de_comm_*, de_reduce)
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        check(self._lib.de_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        check(self._lib.de_comm_init(self._h, ctypes.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def comm_destroy(self):
        check(self._lib.de_comm_destroy(self._h))

    def reduce(self, root=0, comm=None):
        """Sum the HDR buffers of all ranks onto `root` (RCCL, in place, on the context's stream)."""
        check(self._lib.de_reduce(self._h, ctypes.c_void_p(comm) if comm else None, int(root)))

    def reduce_progressive(self, root=0, comm=None):
        """Out-of-place sum onto `root`: this rank keeps accumulating; the root displays the assembled frame (progressive mode)."""
        check(self._lib.de_reduce_progressive(self._h, ctypes.c_void_p(comm) if comm else None, int(root)))

    def reduce_ordered(self, root=0, comm=None, out_of_place=False):
        """Sample partition: gather the ranks' HDR buffers on `root` and add them in rank order (in place, or into the assembled buffer
        the display reads: progressive mode)."""
        check(self._lib.de_reduce_ordered(self._h, ctypes.c_void_p(comm) if comm else None, int(root), 1 if out_of_place else 0))

    def last_reduce_ms(self):
        ms = ctypes.c_float()
        check(self._lib.de_last_reduce_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def debug_ordered_sum(self, parts, root=0, out_of_place=False):
        """The root's half of reduce_ordered on ONE GPU (include/digital_earth_debug.h): `parts` = per-rank partial sums as (W, H, 3) arrays; part `root`
        is loaded into the accumulation buffer, the others where ncclRecv would put them; the library's own launch adds them in rank order.
        Returns the assembled frame as (W, H, 3)."""
        W, H = self.image_res
        dev = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float32).transpose(1, 0, 2) for p in parts]))      # device layout [H][W][3]
        if dev.shape != (len(parts), H, W, 3):
            raise ValueError("every part must have shape (W, H, 3)")
        out = np.empty((H, W, 3), dtype=np.float32)
        check(self._lib.de_debug_ordered_sum(self._h, dev.ctypes.data, len(parts), int(root), 1 if out_of_place else 0, out.ctypes.data))
        return np.ascontiguousarray(out.transpose(1, 0, 2))

    def debug_standin_reduce(self, extra_copies=0):
        """A one-GPU stand-in for the frame's collective, enqueued where reduce() goes (include/digital_earth_debug.h); last_reduce_ms() then
        reports its queue-to-finish latency."""
        check(self._lib.de_debug_standin_reduce(self._h, int(extra_copies)))

    def set_display_source(self, device_ptr):
        """Display / fetch from another [H][W][3] f32 device buffer (None: the accumulation buffer again)."""
        check(self._lib.de_set_display_source(self._h, ctypes.c_void_p(device_ptr) if device_ptr else None))

    def set_launch_slots(self, n_slots, n_big=None):
        """Launches in flight: n_slots for small launches, n_big for big ones (1 = serial)."""
        check(self._lib.de_set_launch_slots(self._h, int(n_slots), int(n_big if n_big is not None else min(n_slots, 3))))

    def set_wave_budget(self, waves_per_cu):
        check(self._lib.de_set_wave_budget(self._h, int(waves_per_cu)))

    def render_to_image_device(self):
        """Run the display transform and leave the (W, H, 3) image on the device; returns its address."""
        if not self._textures_copied:
            self.copy_textures()
        p = ctypes.c_void_p()
        check(self._lib.de_render_to_image(self._h, ctypes.byref(p)))
        return p.value

    def set_stream(self, hip_stream_handle):
        """Use the caller's HIP stream as the context stream; 0 / None is HIP's null stream (torch's default stream)."""
        check(self._lib.de_set_stream(self._h, ctypes.c_void_p(hip_stream_handle or None)))

    def flush(self):
        """Make the context stream wait (on the device) for every accumulate() issued so far — before the caller enqueues
        work of its own on that stream (parallel.reduce_hdr on the bound tensor)."""
        check(self._lib.de_flush(self._h))

    def upload_hdr(self, hdr, spp):
        """Load an accumulation state ((W, H, 3) float32 sums, sample count) — resume of a progressive render."""
        hdr = np.ascontiguousarray(hdr, dtype=np.float32)
        if hdr.shape != (self.image_res[0], self.image_res[1], 3):
            raise ValueError("hdr must have shape (W, H, 3)")
        check(self._lib.de_upload_hdr(self._h, hdr.ctypes.data, int(spp)))
        self.current_spp = int(spp)

    def save_checkpoint(self, path):
        """Persist the progressive state — HDR sums, sample count, RNG base seed and the scalar parameters — so that a long
        render (BASELINE cfg5: 1024 spp) can be resumed.  The reference keeps this state in memory only (renderer.py:23,25)."""
        np.savez(path, hdr=self.fetch_hdr(), spp=np.int64(self.current_spp), seed=np.uint64(self.seed),
                 params=np.frombuffer(bytes(self._params), dtype=np.uint8), image_res=np.array(self.image_res))

    def load_checkpoint(self, path):
        """Resume from save_checkpoint(): the next accumulate() continues with sample index `spp`, bit-identically."""
        z = np.load(path)
        if tuple(int(x) for x in z["image_res"]) != self.image_res:
            raise ValueError("checkpoint is %s, renderer is %s" % (tuple(z["image_res"]), self.image_res))
        ctypes.memmove(ctypes.byref(self._params), z["params"].tobytes(), ctypes.sizeof(self._params))
        self._push_params()
        self.seed = int(z["seed"])
        self.upload_hdr(z["hdr"], int(z["spp"]))

    def set_current_spp(self, spp):
        self.current_spp = int(spp)
        check(self._lib.de_set_current_spp(self._h, int(spp)))

    def debug_samples(self, sample_index):
        """Per pixel [H][W]: radiance, wavelength, RNG draws, vertices of sample `sample_index` (not accumulated)."""
        if not self._textures_copied:
            self.copy_textures()
        out = np.empty((self.image_res[1], self.image_res[0], 4), dtype=np.float32)
        check(self._lib.de_debug_samples(self._h, self.seed, int(sample_index), out.ctypes.data))
        return out

    def debug_math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            bp = b.ctypes.data
        check(self._lib.de_debug_math(self._h, int(fn), a.ctypes.data, bp, out.ctypes.data, a.size))
        return out

    @staticmethod
    def to_vec3u(c):                                                     # renderer.py:386-393
        return (np.clip(np.asarray(c, dtype=np.float32), 0.0, 1.0) * 255).astype(np.uint8)

    @staticmethod
    def to_vec3(c):                                                      # renderer.py:395-401
        return np.asarray(c, dtype=np.float32) / 255.0
