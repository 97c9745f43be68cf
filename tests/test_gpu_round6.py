"""Round 6: the device code of the N-rank collectives on one GPU (ordered sum with several parts, the stand-in collective), the window loop
pipelined (de_fetch_image_begin / _end), CU-masked render streams, a borrower that owns one map, an empty share's timing."""
import numpy as np
import pytest

from helpers import make_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Renderer():
    from digital_earth_amd.renderer import Renderer as R
    return R


def _small(Renderer, W=128, H=64, seed=21, **kw):
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=seed, **kw)
    r.set_fov(0.42)
    r.copy_textures()
    return r


def _parts(n_parts, shape, seed=5):
    rng = np.random.default_rng(seed)
    # magnitudes seven decades apart, signs mixed: any other association of the additions shows in the last bits
    return [(rng.standard_normal(shape) * 10.0 ** rng.integers(-3, 4, size=shape)).astype(np.float32) for _ in range(n_parts)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_parts,root,out_of_place", [(2, 0, False), (3, 2, False), (4, 0, True), (4, 1, False), (8, 0, False), (8, 5, True)])
def test_ordered_sum_kernel_with_several_parts(Renderer, n_parts, root, out_of_place):
    """de_reduce_ordered's root half — the gather buffer's per-rank offsets, the root's part read IN PLACE, ordered_sum_kernel with
    n_parts > 1 — had never executed: RCCL refuses two ranks on one device.  de_debug_ordered_sum feeds the same launch from the host.  Result =
    ((p0 + p1) + p2) + ... in f32, bit for bit, wherever the root sits."""
    W, H = 80, 24
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(256, 128))
    parts = _parts(n_parts, (W, H, 3))
    got = r.debug_ordered_sum(parts, root=root, out_of_place=out_of_place)
    want = parts[0].copy()
    for p in parts[1:]:
        want = want + p
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # a plain numpy sum in another order is NOT the same bits (the test would pass vacuously with benign data)
    other = parts[-1].copy()
    for p in parts[-2::-1]:
        other = other + p
    assert n_parts == 2 or (other.view(np.uint32) != want.view(np.uint32)).any()
    # in place: the accumulation buffer now holds the assembled frame; out of place: the display source does, the accumulation buffer keeps the root's part
    assert (r.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all()
    if out_of_place:
        r.set_display_source(None)
        assert (r.fetch_hdr().view(np.uint32) == parts[root].view(np.uint32)).all()
    r.close()


@pytest.mark.timeout(600)
def test_ordered_sum_of_rendered_sample_partitions_equals_the_host_assembly(Renderer):
    """Four ranks' shares of a sample partition rendered here, assembled by the library's launch: the same bits as the host-side rank-ordered sum
    (what parallel.reduce_hdr_ordered computes over gloo)."""
    WORLD = 4
    parts = []
    for rank in range(WORLD):
        r = _small(Renderer)
        r.set_sample_partition(rank, WORLD)
        r.accumulate(8)
        parts.append(r.fetch_hdr())
        r.close()
    want = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    r = _small(Renderer)
    got = r.debug_ordered_sum(parts, root=0)
    assert got.max() > 0 and (got.view(np.uint32) == want.view(np.uint32)).all()
    r.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("lag", [1, 2, 3])
def test_pipelined_window_loop_equals_the_synchronous_one_shifted_by_a_frame(Renderer, lag):
    """accumulate(1) + fetch_image(lag=L) per frame returns frame k - L while frame k + 1 renders; the sequence of images equals the
    synchronous loop's, L calls later, bit for bit — through a reset in the middle (a moved camera) too."""
    W, H, N = 256, 128, 7
    a = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=3)
    a.set_fov(0.42); a.copy_textures()
    sync = []
    for k in range(N):
        if k == 4:
            a.set_fov(0.3); a.reset_framebuffer()
        a.accumulate(1)
        sync.append(a.fetch_image())
    b = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=3)
    b.set_fov(0.42); b.share_textures_from(a)
    lagged = []
    for k in range(N):
        if k == 4:
            b.set_fov(0.3); b.reset_framebuffer()
        b.accumulate(1)
        lagged.append(b.fetch_image(lag=lag))
    assert all(x is None for x in lagged[:lag]) and all(x is not None for x in lagged[lag:])
    with pytest.raises(RuntimeError):
        b.fetch_image()                       # a synchronous fetch while lagged ones are in flight would hand out frames out of order
    rest = b.fetch_pending(all_images=True)
    assert len(rest) == lag and b.fetch_pending() is None
    got = lagged[lag:] + rest
    for k in range(N):
        assert got[k].shape == (W, H, 3) and (got[k].view(np.uint32) == sync[k].view(np.uint32)).all(), k
    # zero-copy views of the pipelined loop: valid until the next fetch, and close() refuses while one is referenced
    for _ in range(lag):
        b.accumulate(1); assert b.fetch_image(lag=lag, copy=False) is None
    b.accumulate(1); v = b.fetch_image(lag=lag, copy=False)
    a.accumulate(1); want = a.fetch_image()
    assert not v.flags.writeable and (np.asarray(v).view(np.uint32) == want.view(np.uint32)).all()
    with pytest.raises(RuntimeError):
        b.close()
    del v
    b.fetch_pending(); b.close(); a.close()


@pytest.mark.timeout(600)
def test_five_fetches_in_flight_are_refused(Renderer):
    from digital_earth_amd import _native
    r = _small(Renderer)
    r.accumulate(1)
    for _ in range(4):
        _native.check(r._lib.de_fetch_image_begin(r._h))
    assert r._lib.de_fetch_image_begin(r._h) == -4            # DE_ERR_STATE: four fetches in flight is the ring's size
    r._fetches = 4
    assert r.fetch_pending() is not None
    import ctypes
    ptr = ctypes.POINTER(ctypes.c_float)()
    assert r._lib.de_fetch_image_end(r._h, ctypes.byref(ptr)) == -4
    r.close()


@pytest.mark.timeout(900)
def test_cu_masked_render_streams_give_the_same_frame(Renderer, lut_arrays):
    """de_tuning.v6_cu_withhold: the launch slots' streams leave CUs of every XCD to the context stream's small kernels and the persistent grid shrinks
    with the mask.  Where a path runs never shows in its bits: same frame as the oracle's, also with launches overlapping and the tail chain exporting."""
    W, H = 256, 128
    r = Renderer((W, H), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), seed=9)
    r.set_fov(0.42); r.copy_textures()
    o = make_oracle(W, H, lut_arrays, [r.download_texture(s) for s in range(7)], r._params)
    o.accumulate(6, 9)
    want = o.fetch_hdr()
    for withhold in (1, 4, 0):
        t = r.tuning(); t.v6_cu_withhold = withhold; t.v6_tail_when_alone = 1; t.v6_tail_min_paths = 4096; r.set_tuning(t)
        assert r.tuning().v6_cu_withhold == withhold
        r.reset_framebuffer()
        for n in (1, 2, 3):
            r.accumulate(n)
        assert r.last_call_info()["variant"] == 6
        assert (r.fetch_hdr().view(np.uint32) == want.view(np.uint32)).all(), withhold
    t = r.tuning(); t.v6_cu_withhold = 32
    with pytest.raises(Exception):
        r.set_tuning(t)
    r.close()


@pytest.mark.timeout(600)
def test_standin_collective_leaves_the_frame_alone(Renderer):
    r = _small(Renderer)
    r.accumulate(3)
    before = r.fetch_hdr()
    for extra in (0, 7):
        r.debug_standin_reduce(extra)
        assert r.last_reduce_ms() >= 0.0
    r.accumulate(2)
    r.debug_standin_reduce(0)
    r2 = _small(Renderer)
    r2.accumulate(3); r2.accumulate(2)
    assert (r2.fetch_hdr().view(np.uint32) == r.fetch_hdr().view(np.uint32)).all() and (before != 0).any()
    r.close(); r2.close()


@pytest.mark.timeout(600)
def test_a_borrower_that_owns_its_cloud_map(Renderer, lut_arrays):
    """bench.py's cfg4 leg: the maps of the resident Earth borrowed, the cloud-heavy cloud map its own (Renderer.copy_texture after share_textures_from).
    The frame equals the oracle's on exactly those seven maps, and the lender's frame is unchanged."""
    from digital_earth_amd import textures as T
    a = _small(Renderer)
    b = Renderer((128, 64), (0, 1, 0), texture_source="synthetic", texture_size=(2048, 1024), cloud_heavy=True, seed=21)
    b.set_fov(0.42)
    b.share_textures_from(a)
    b.copy_texture(T.CLOUDS)
    maps = [b.download_texture(s) for s in range(7)]
    assert (maps[T.CLOUDS] != a.download_texture(T.CLOUDS)).any() and (maps[T.TOPOGRAPHY] == a.download_texture(T.TOPOGRAPHY)).all()
    b.accumulate(4)
    o = make_oracle(128, 64, lut_arrays, maps, b._params)
    o.accumulate(4, 21)
    assert (b.fetch_hdr().view(np.uint32) == o.fetch_hdr().view(np.uint32)).all()
    a.accumulate(4)
    own = _small(Renderer); own.accumulate(4)
    assert (a.fetch_hdr().view(np.uint32) == own.fetch_hdr().view(np.uint32)).all()
    b.close(); a.close(); own.close()


@pytest.mark.timeout(600)
def test_an_empty_share_reports_no_duration(Renderer):
    """a call whose share of a sample partition is empty launches nothing; de_last_accumulate_ms answers 0 for it, not the previous call's time."""
    r = _small(Renderer)
    r.set_sample_partition(3, 4)
    r.accumulate(4)                       # index 3: one launch
    assert r.last_accumulate_ms() > 0.0
    r.accumulate(2)                       # indices 4, 5: none = 3 (mod 4)
    assert r.last_accumulate_ms() == 0.0
    r.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("lag", [1, 2])
def test_earth_viewer_frame_loop_pipelined(lag):
    """EarthViewer.frame(pipelined=L): the reference's window loop (earth_viewer.py:203-317) with display and host copy of an iteration running while the next renders.
    The images are those of the synchronous loop, L iterations later — through a slider that resets the framebuffer (sun angle), one that does not (exposure) and a
    camera move; finish() hands out the rest."""
    from digital_earth_amd.earth_viewer import EarthViewer
    kw = dict(screen_res=(256, 128), texture_source="synthetic", texture_size=(1024, 512), seed=5)

    def script(v, k):
        if k == 3:
            return dict(sun_angle=0.9)
        if k == 5:
            return dict(exposure=1.7)
        if k == 6:
            v.camera.rotate(0.01, 0.0)
        return {}
    a = EarthViewer(**kw)
    sync = [a.frame(spp=1, **script(a, k)).copy() for k in range(9)]
    b = EarthViewer(**kw)
    got = [b.frame(spp=1, pipelined=lag, **script(b, k)) for k in range(9)]
    assert all(x is None for x in got[:lag])
    tail = b.renderer.fetch_pending(all_images=True)
    assert len(tail) == lag
    seq = [np.array(x) for x in got[lag:]] + tail
    assert len(seq) == 9
    for k in range(9):
        assert (seq[k].view(np.uint32) == sync[k].view(np.uint32)).all(), k
    assert sync[3].std() > 0 and not (sync[4] == sync[3]).all()      # the sun moved after iteration 3: iteration 4 starts a new accumulation
    assert b.finish() is None
    a.close(); b.close()
