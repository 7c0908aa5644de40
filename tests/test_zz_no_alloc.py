"""GPU test of ptamd_render_counted_noalloc, the entry point for callers that cannot allocate: through the C ABI, its
film and its ten counters must equal those of ptamd_render with PTAMD_FLAG_COUNT + ptamd_render_stats bit for bit
(integer counters, and a film whose per-sample values do not depend on the entry point), on two calls in a row, the
second with a smaller input (fewer samples on a sub-rectangle), which a counter array that kept the first call's
contents would fail."""
import ctypes as C

import numpy as np
import pytest

from scenes import presets

pytestmark = pytest.mark.gpu


def test_counted_noalloc_matches_render_with_count(gpu):
    import torch
    scene = presets.mixed("tests/golden/blob.glb", 160, 96, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
    pt, cam = gpu.build_scene(scene)
    w, h = scene["film"]
    L = gpu.lib()
    d_stats = torch.full((10,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    names = [n for n, _ in gpu.CStats._fields_]
    for spp_begin, spp, depth, rect in [(0, 6, 6, None), (6, 2, 4, (24, 16, 101, 77))]:
        ref_film, film = gpu.Film(w, h), gpu.Film(w, h)
        gpu.Render(cam, pt, ref_film, spp, depth, spp_begin=spp_begin, seed=41, rect=rect, count=True)
        want = gpu.RenderStats()
        r = gpu.CRect(*rect) if rect is not None else None
        gpu.check(L.ptamd_render_counted_noalloc(pt._h, C.byref(cam.to_c()), film._h, spp_begin, spp, depth, 0, 41,
                                                 C.byref(r) if r is not None else None, 0, C.c_void_p(d_stats.data_ptr())))
        gpu.synchronize()
        got = dict(zip(names, d_stats.cpu().tolist()))
        assert got == want, (got, want)
        assert want["samples"] == (w * h if rect is None else (rect[2] - rect[0]) * (rect[3] - rect[1])) * spp and want["nodes"] > 0
        assert np.array_equal(film.Accum().view(np.uint32), ref_film.Accum().view(np.uint32))
        assert np.array_equal(film.Samples(), ref_film.Samples())
    # an empty request still clears the counters and reports zeros; a null output pointer is refused
    film = gpu.Film(w, h)
    gpu.check(L.ptamd_render_counted_noalloc(pt._h, C.byref(cam.to_c()), film._h, 0, 0, 4, 0, 41, None, 0, C.c_void_p(d_stats.data_ptr())))
    gpu.synchronize()
    assert d_stats.cpu().tolist() == [0] * 10
    assert L.ptamd_render_counted_noalloc(pt._h, C.byref(cam.to_c()), film._h, 0, 1, 4, 0, 41, None, 0, None) < 0
