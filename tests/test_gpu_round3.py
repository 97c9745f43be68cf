"""GPU tests of the round-3 host-side changes: launch slots (consecutive accumulate calls overlap, sums stay in call order),
the null-stream meaning of de_set_stream(NULL), loans recorded by de_share_textures, the RCCL loader's failure path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import bits_equal_fraction

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


KW = dict(texture_source="synthetic", texture_size=(1024, 512))


def test_many_small_launches_equal_one_big_launch(Renderer):
    """k x accumulate(1) — spread over the launch slots, running concurrently — gives the HDR sums of accumulate(k) bit for bit,
    also with resets, fetches and parameter changes in between (each orders itself against the launches in flight)."""
    a = Renderer((128, 64), (0, 1, 0), seed=5, **KW); a.set_fov(0.45)
    a.accumulate(24)
    want = a.fetch_hdr()
    b = Renderer((128, 64), (0, 1, 0), seed=5, **KW); b.set_fov(0.45)
    for _ in range(24):
        b.accumulate(1)
    assert bits_equal_fraction(b.fetch_hdr(), want) == 1.0
    # reset in the middle of a burst, then the same burst again: nothing of the first burst may leak into the second
    b.reset_framebuffer()
    for _ in range(7):
        b.accumulate(1)
    b.reset_framebuffer()
    for k in range(24):
        b.accumulate(1)
        if k == 11:
            mid = b.fetch_hdr()               # a fetch between launches sees exactly the launches issued so far
    assert bits_equal_fraction(b.fetch_hdr(), want) == 1.0
    a.reset_framebuffer(); a.accumulate(12)
    assert bits_equal_fraction(mid, a.fetch_hdr()) == 1.0
    # a camera change between launches rebuilds the frame constants: launches in flight keep the old ones
    a.reset_framebuffer(); a.accumulate(3); a.set_fov(0.3); a.accumulate(3)
    want2 = a.fetch_hdr()
    b.set_fov(0.45); b.reset_framebuffer()
    for _ in range(3):
        b.accumulate(1)
    b.set_fov(0.3)
    for _ in range(3):
        b.accumulate(1)
    assert bits_equal_fraction(b.fetch_hdr(), want2) == 1.0


@pytest.mark.parametrize("slots", ["1", "2", "8"])
def test_slot_count_does_not_change_the_image(slots):
    """DE_SLOTS is read at de_create: run the same burst in child processes with 1 (serial), 2 and 8 slots."""
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from digital_earth_amd.renderer import Renderer\n"
            "r = Renderer((64, 64), (0, 1, 0), seed=3, texture_source='synthetic', texture_size=(512, 256)); r.set_fov(0.4)\n"
            "[r.accumulate(1) for _ in range(10)]; r.accumulate(3)\n"
            "np.save(sys.argv[1], r.fetch_hdr())\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "hdr.npy")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, DE_SLOTS=slots))
        got = np.load(out)
    from digital_earth_amd.renderer import Renderer
    r = Renderer((64, 64), (0, 1, 0), seed=3, texture_source="synthetic", texture_size=(512, 256)); r.set_fov(0.4)
    r.accumulate(13)
    assert bits_equal_fraction(got, r.fetch_hdr()) == 1.0


def test_distributed_frame_on_the_default_stream():
    """parallel.DistributedFrame(stream=None) takes torch's current stream — the null stream, handle 0.  de_set_stream(NULL)
    must mean exactly that stream: the zero-fill, the launches, torch's reduce and the display are then ordered.
    Child process with torch imported first, like bench.py (torch brings its own HIP runtime; it has to be the one loaded)."""
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "torch.cuda.set_device(0)\n"
            "from digital_earth_amd.renderer import Renderer\n"
            "from digital_earth_amd import parallel\n"
            "from helpers import bits_equal_fraction\n"
            "KW = dict(texture_source='synthetic', texture_size=(1024, 512))\n"
            "a = Renderer((64, 32), (0, 1, 0), seed=4, **KW); a.set_fov(0.4)\n"
            "a.accumulate(3)\n"
            "want_hdr, want_img = a.fetch_hdr(), a.fetch_image()\n"
            "b = Renderer((64, 32), (0, 1, 0), seed=4, **KW); b.set_fov(0.4)\n"
            "f = parallel.DistributedFrame(b, 0, 1, 0, stream=None)\n"
            "assert f.stream.cuda_stream == 0\n"
            "for _ in range(3):\n"
            "    f.render_frame(3)\n"
            "    b.flush()\n"
            "    total = float(f.hdr.sum().item())      # torch's own work on the null stream sees the finished HDR buffer\n"
            "    ref = float(want_hdr.astype(np.float64).sum())\n"
            "    assert abs(total - ref) <= 1e-3 * abs(ref), (total, ref)\n"
            "assert bits_equal_fraction(f.fetch_hdr(), want_hdr) == 1.0\n"
            "assert (b.fetch_image() == want_img).all()\n"
            "f.close()\n"
            "b.accumulate(1)\n"
            "assert np.isfinite(b.fetch_hdr()).all()\n"
            "print('default stream ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    out = subprocess.check_output([sys.executable, "-c", code])
    assert b"default stream ok" in out


def test_lent_maps_cannot_be_freed_or_repacked(Renderer):
    """de_share_textures records the loan on the lender: while it lasts the lender refuses to be destroyed, to
    replace or repack its maps and to replace its LUTs; the borrower cannot download what it does not own through the C ABI."""
    import ctypes
    from digital_earth_amd import _native
    from digital_earth_amd._native import DigitalEarthError, DE_FLAG_CLAMP_SAMPLER
    owner = Renderer((64, 32), (0, 1, 0), seed=1, **KW); owner.set_fov(0.4)
    owner.copy_textures()
    owner.accumulate(1)
    want = owner.fetch_hdr()
    user = Renderer((64, 32), (0, 1, 0), seed=1, **KW); user.set_fov(0.4)
    user.share_textures_from(owner)
    user.accumulate(1)
    assert bits_equal_fraction(user.fetch_hdr(), want) == 1.0
    with pytest.raises(DigitalEarthError) as e:
        owner.close()
    assert e.value.code == -4 and "borrow" in str(e.value)
    with pytest.raises(DigitalEarthError):
        owner.copy_textures()                # would free the maps under the borrower
    owner.set_flag(DE_FLAG_CLAMP_SAMPLER, True)
    with pytest.raises(DigitalEarthError) as e:
        owner.accumulate(1)                  # would repack them in place for another address mode
    assert "lent" in str(e.value)
    owner.set_flag(DE_FLAG_CLAMP_SAMPLER, False)
    owner.trim_textures()                    # allowed: the borrower never held the as-uploaded copies
    buf = np.zeros((512, 1024, 1), np.uint8)
    rc = _native.load().de_download_texture(user._h, 1, buf.ctypes.data, buf.nbytes)
    assert rc == -4
    user.accumulate(1); owner.accumulate(1)
    user.close()                             # the loan ends with the borrower
    owner.copy_textures()
    owner.close()


def test_borrower_with_maps_of_its_own_ends_the_loan(Renderer):
    owner = Renderer((64, 32), (0, 1, 0), **KW); owner.copy_textures()
    user = Renderer((64, 32), (0, 1, 0), **KW)
    user.share_textures_from(owner)
    user.copy_textures()                     # uploads / generates all seven maps and the LUTs again
    user._lender = None
    owner.close()                            # no loan left
    user.accumulate(1)
    assert np.isfinite(user.fetch_hdr()).all()


def test_rccl_loader_failure_is_an_error_not_a_crash():
    """DE_RCCL_PATH pointing at a missing file and no search beyond it: de_comm_unique_id returns DE_ERR_STATE with a message
    (the old path called dlerror() twice and built a std::string from NULL).  Child process: the loader state is global."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from digital_earth_amd.renderer import Renderer\n"
            "from digital_earth_amd._native import DigitalEarthError\n"
            "r = Renderer((64, 32), (0, 1, 0), texture_source='constant')\n"
            "try:\n"
            "    r.comm_unique_id()\n"
            "except DigitalEarthError as e:\n"
            "    assert e.code == -4 and 'cannot load librccl' in str(e), str(e)\n"
            "    print('clean error')\n" % ROOT)
    out = subprocess.check_output([sys.executable, "-c", code],
                                  env=dict(os.environ, DE_RCCL_PATH="/nonexistent/librccl.so", DE_RCCL_STRICT="1"))
    assert b"clean error" in out


def test_cfg3_quality2_maps_full_sample_count(Renderer, lut_arrays):
    """BASELINE configs[2] with the map tier it names (the full-size preset tests used 4096x2048 maps and compared
    the first 2 spp): 1920x1080, `config - florida.txt`, quality-2 maps (21600x10800 / 16200x8100), all 256 spp.  The oracle
    renders every 97th 8x8 tile of the same frame at the full sample count; those pixels match bit for bit."""
    from digital_earth_amd import parallel
    from helpers import make_oracle, load_preset
    W, H, SPP = 1920, 1080, 256
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_quality=2, seed=2)
    r.copy_textures()
    p = load_preset("config - florida.txt")
    r.set_camera_pos(*p["pos"]); r.set_look_at(*p["look_at"]); r.set_up(*p["up"])
    r.set_fov(p["fov"]); r.set_aspect_scale(p["aspect_scale"]); r.set_exposure(p["exposure"])
    r.set_crf(p["crf"]); r.set_gamma(p["gamma"]); r.set_sun_angle(p["sun_angle"]); r.set_sun_path_rot(p["sun_path_rot"])
    for _ in range(4):
        r.accumulate(64)                      # four launches of 64 spp, overlapping through the launch slots
    g = r.fetch_hdr()
    assert r.current_spp == SPP and np.isfinite(g).all() and g.max() > 0
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(SPP, 2, 11, 97)
    mask = parallel.owned_pixel_mask(W, H, 11, 97)
    assert mask.sum() >= 20000
    assert (o.fetch_hdr()[mask].view(np.uint32) == g[mask].view(np.uint32)).all()
    img = r.fetch_image()
    assert img.min() >= 0.0 and img.max() <= 1.0 and img.std() > 0.01
