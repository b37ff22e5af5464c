"""HipPathTracer: the Python face of the device integrator.

Mirrors the shape of the reference's ``Renderer`` (/root/reference/src/
GoblinRenderer.h:53-60): construct from the render settings, ``render(scene)``
fills the Film.  All computation happens in libgoblin_hip.so through the C ABI
(include/goblin_hip.h); PyTorch only supplies device memory and streams.  There
is no CPU fallback: without the HIP library or a GPU this raises.
"""
import ctypes as C

import numpy as np

from . import _abi


def _torch():
    import torch
    return torch


class Film:
    """Accumulators {sum w*L.rgb, sum w} as a (yres, xres, 4) float32 CUDA tensor
    (the reference's Pixel array, GoblinFilm.h:16-23)."""

    def __init__(self, xres, yres, device):
        torch = _torch()
        self.xres, self.yres = xres, yres
        self.accum = torch.zeros((yres, xres, 4), dtype=torch.float32, device=device)

    def zero_(self):
        self.accum.zero_()

    def normalized(self):
        """Film::writeImage's rgb / weight (GoblinFilm.cpp:164-172) as a tensor."""
        torch = _torch()
        w = self.accum[..., 3:4]
        rgb = self.accum[..., :3] * (1.0 / w)
        return torch.where(w > 0, rgb, torch.zeros_like(rgb))

    def numpy(self):
        return self.accum.detach().cpu().numpy()


class HipPathTracer:
    """Device path tracer / AO renderer bound to one scene on one GPU."""

    def __init__(self, scene, device=0, bvh="host"):
        """bvh: "host" (binned SAH, the default) or "device" (Morton-sorted linear BVH built on the GPU)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("HipPathTracer needs a HIP device: torch.cuda.is_available() is False and there is no "
                               "CPU fallback on the product path")
        self.lib = _abi.hip_lib()
        self.scene = scene
        self.device_index = device if isinstance(device, int) else torch.device(device).index or 0
        self.device = torch.device("cuda", self.device_index)
        handle = C.c_void_p()
        if bvh not in ("host", "device"):
            raise ValueError("bvh must be 'host' or 'device'")
        if bvh == "device":
            st = self.lib.gbl_create_ex(scene.desc_ptr, self.device_index, _abi.GBL_CREATE_DEVICE_BVH, C.byref(handle))
        else:
            st = self.lib.gbl_create(scene.desc_ptr, self.device_index, C.byref(handle))
        if st != _abi.GBL_OK:
            raise _abi.GoblinError(st, self.lib.gbl_last_error(None).decode())
        self.handle = handle
        info = _abi.gbl_info()
        self.lib.gbl_get_info(self.handle, C.byref(info))
        self.info = info
        self.window = tuple(info.window)

    def update_instances(self, first, transforms):
        """Move instances first.. to new (position, orientation wxyz, scale) transforms and rebuild the TLAS in place."""
        arr = (_abi.gbl_trs * len(transforms))()
        for i, (pos, quat, scale) in enumerate(transforms):
            arr[i].position[:] = pos
            arr[i].orientation[:] = quat
            arr[i].scale[:] = scale
        st = self.lib.gbl_update_instances(self.handle, first, len(transforms), arr)
        if st != _abi.GBL_OK:
            raise _abi.GoblinError(st, self.lib.gbl_last_error(self.handle).decode())
        self.lib.gbl_get_info(self.handle, C.byref(self.info))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.lib.gbl_destroy(h)
            except Exception:
                pass

    def timings(self, n=1):
        """Device times (ms) of the last n render() calls, most recent first: list of (main_kernel_ms, total_ms)."""
        buf = (_abi.gbl_timing * n)()
        got = self.lib.gbl_get_timings(self.handle, n, buf)
        return [(buf[i].main_kernel_ms, buf[i].total_ms) for i in range(got)]

    def valu_issue(self, op, waves_per_simd, iters=4096):
        """gbl_selftest_valu_issue: {ms, wave_instructions, ticks_per_wave, ticks_per_instruction, realtime_ticks_per_wave, clock_ghz}
        of one launch made after two seconds of the same launch back to back."""
        out = (C.c_double * 8)()
        st = self.lib.gbl_selftest_valu_issue(self.handle, int(op), int(waves_per_simd), int(iters), out)
        if st != _abi.GBL_OK:
            raise _abi.GoblinError(st, self.lib.gbl_last_error(self.handle).decode())
        return {"ms": out[0], "wave_instructions": out[1], "ticks_per_wave": out[2], "ticks_per_instruction": out[3],
                "realtime_ticks_per_wave": out[4], "clock_ghz": out[5],
                "longest_span_us": out[6] * 0.01, "shortest_span_us": out[7] * 0.01}

    def new_film(self):
        return Film(self.info.xres, self.info.yres, self.device)

    def _params(self, setting=None, window=None, seed=0, replay=None, li_out=None, stats=False, rr=False, shard=None,
                schedule=0, sampler="native", exact_ties=False):
        s = setting or self.scene.desc.setting
        p = _abi.gbl_render_params()
        p.integrator = s.integrator
        p.sample_per_pixel = s.sample_per_pixel
        p.max_ray_depth = s.max_ray_depth
        p.ao_sample_num = s.ao_sample_num
        p.bssrdf_sample_num = s.bssrdf_sample_num
        w = window or (0, 0, 0, 0)
        for i in range(4):
            p.window[i] = int(w[i])
        if shard is not None:
            p.tile_shard_index, p.tile_shard_count = int(shard[0]), int(shard[1])
        p.sample_mode = _abi.GBL_SAMPLES_REPLAY if replay is not None else (
            _abi.GBL_SAMPLES_STREAM if sampler == "stream" else _abi.GBL_SAMPLES_NATIVE)
        p.seed = int(seed)
        p.replay_samples = replay.data_ptr() if replay is not None else None
        p.li_out = li_out.data_ptr() if li_out is not None else None
        p.russian_roulette = 1 if rr else 0
        p.collect_stats = 1 if stats else 0
        p.schedule = {"auto": 0, "megakernel": 1, "wavefront": 2}.get(schedule, schedule)
        p.exact_ties = 1 if exact_ties else 0
        torch = _torch()
        p.stream = torch.cuda.current_stream(self.device).cuda_stream
        return p

    def render(self, film=None, setting=None, window=None, seed=0, replay_samples=None, want_li=False, stats=False,
               timed=False, rr=False, shard=None, schedule="auto", sampler="native", exact_ties=False):
        """Accumulate one pass into ``film`` (created if None).

        replay_samples: (n, dims) float32 tensor/array of Sample records for the
        window, pixel-major (GBL_SAMPLES_REPLAY); otherwise the native sampler.
        sampler: "native" (counter-based law) or "stream" (the reference's own mt19937 stream, GBL_SAMPLES_STREAM).
        shard: (index, count) renders only every count-th 8x8 sample tile (multi-GPU).
        exact_ties: native sampler on lean scenes -- keep the reference's exact-t tie rule (gbl_render_params.exact_ties).
        Returns dict(film=..., li=..., stats=...).
        """
        torch = _torch()
        if film is None:
            film = self.new_film()
        replay = None
        if replay_samples is not None:
            replay = torch.as_tensor(np.ascontiguousarray(replay_samples, np.float32) if isinstance(
                replay_samples, np.ndarray) else replay_samples, dtype=torch.float32, device=self.device).contiguous()
        s = setting or self.scene.desc.setting
        w = window or self.window
        spp = _abi.host_lib().gbl_host_round_to_square(s.sample_per_pixel)
        npaths = (w[1] - w[0]) * (w[3] - w[2]) * spp
        if replay is not None:
            dims = _abi.host_lib().gbl_host_sample_dimension_scene(C.byref(self.scene.desc), C.byref(s))
            if tuple(replay.shape) != (npaths, dims):
                raise ValueError("replay_samples must have shape (%d, %d), got %s" % (npaths, dims, tuple(replay.shape)))
        li = torch.zeros((npaths, 4), dtype=torch.float32, device=self.device) if want_li else None
        p = self._params(s, window, seed, replay, li, stats, rr, shard, schedule, sampler, exact_ties)
        st_out = _abi.gbl_stats() if (stats or timed) else None
        st = self.lib.gbl_render(self.handle, C.byref(p), film.accum.data_ptr(), C.byref(st_out) if st_out else None)
        if st != _abi.GBL_OK:
            raise _abi.GoblinError(st, self.lib.gbl_last_error(self.handle).decode())
        return {"film": film, "li": li, "stats": st_out.as_dict() if st_out else None, "paths": npaths}
