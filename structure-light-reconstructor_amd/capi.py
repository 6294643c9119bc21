"""ctypes binding of libslr_hip.so (include/slr.h) -- the only compute path of this package.

numpy arrays are passed as SLR_MEM_HOST, torch CUDA tensors as SLR_MEM_DEVICE.  There is no CPU fallback:
if the shared library is missing or no GPU is usable, an exception is raised (never a silent fallback).
PyTorch is used only for device memory, streams and torch.distributed -- plumbing, not the product.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslr_hip.so")

MF_PLANES = 14
MAX_GRAY_BITS = 16
MEM_HOST, MEM_DEVICE = 0, 1
OPT_MF_MATCH_ALGO = 1          # 0 auto, 1 linear sweep, 2 indexed (radix-sorted distinct phases), 3 indexed (counting sort), 4/5/6 lean shapes (1024x4, 512x8 two rows per CU, 512x8 three rows per CU), 7 persistent grouped K4 (FORMS=all builds only), 8 = the lean kernel auto picks, 9 = its index without the hash dedup (FORMS=all builds only)
OPT_RECT_DECODE_ALGO = 3       # fused rectify+decode: 0 auto (5, else 6), 1 direct gather, 2 LDS tiles 64x16, 3 sliding LDS
                               # window down tile columns, 4 tiles 128x8 / 256 threads, 5 tiles 128x8 / 512 threads,
                               # 6 tiles 64x8 / 256 threads
OPT_RECT_DMA_SHAPE = 6         # tile / threads of form 7 (LDS-DMA fused decode): 0 = 256x16/512, 1 = 256x8/512, 2 = 256x8/256,
                               # 3 = 128x16/512, 4 = 128x8/256, 5 = 256x4/256, 6 = 128x16/256
OPT_RECT_DMA_DEPTH = 7         # phases of LDS-DMA in flight ahead of the decode: 1 or 2
OPT_DEBUG_RECT_RESIDENT = 8    # tests: workgroups of the persistent fused decodes (0 = resident set)
OPT_DEBUG_FLAGS = 9            # tests: bit 0 no map digest, bit 1 no buffer-descriptor form, bit 2 no lean K5, bit 3 no fused decode+count
OPT_DEBUG_K4_STOP = 10         # -DSLR_DEBUG_HOOKS builds only
OPT_BATCH_STREAMS = 12         # GRAY_ONLY batch: 2 (default) = frames pipelined over two streams, 1 = sequential
OPT_DEBUG_POISON_SCRATCH = 13  # tests: scratch buffers are filled with 0x7B bytes before a call gets them
OPT_MF_BATCH_GROUP = 15        # frames per match launch of reconstruct_mf_batch (default 8; 1 = frame by frame)
OPT_MF_BATCH_DECODE_GROUP = 16 # frames per fused-decode launch inside such a group (1..8, default 8)
OPT_EVAL_MODEL = 14            # 0 = strict IEEE (default), 1 = the reference's MSVC2010 x87 / fp:precise evaluation (slr.h)
OPT_HYBRID_ONE_PASS = 11       # hybrid stacks: 0 = two fused launches (Gray planes, then white/black + fringes), 1 = one kernel
OPT_PROFILE_STRIDE = 5         # the HIP-event profiler brackets every n-th launch of a kernel
OPT_ASYNC_HOST = 4             # host-buffer calls return after enqueuing; outputs valid after ctx.synchronize()
OPT_MF_DECODE_VEC = 2          # 0 auto, 4 / 8 / 16 pixels per thread in the unfused K2 kernel

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_CONFIGURED, ERR_UNSUPPORTED, ERR_OOM = 0, -1, -2, -3, -4, -5, -6

# every symbol include/slr.h declares (checked by tests/test_capi_symbols.py without a GPU)
SYMBOLS = [
    "slr_version", "slr_status_string", "slr_current_device", "slr_create", "slr_destroy", "slr_set_stream", "slr_synchronize",
    "slr_last_error", "slr_set_option", "slr_set_calibration", "slr_set_rectify_maps", "slr_init_rectify_maps",
    "slr_get_rectify_maps", "slr_get_rectify_info", "slr_remap_u8", "slr_mf_decode", "slr_mfn_decode", "slr_mfn_rectify_decode",
    "slr_rectify_source_rows",
    "slr_mf_rectify_decode", "slr_mf_rectify_decode_pair", "slr_gray_decode", "slr_gray_rectify_decode", "slr_mf_triangulate",
    "slr_mf_triangulate_rows",
    "slr_ge_triangulate", "slr_ray_triangulate", "slr_line_line_intersections", "slr_pointcloud_from_grid", "slr_pointcloud_get",
    "slr_reconstruct_mf", "slr_reconstruct_ge", "slr_reconstruct_gray", "slr_reconstruct_mf_batch", "slr_reconstruct_batch", "slr_reconstruct_mf_cloud", "slr_reconstruct_mf_multi", "slr_reconstruct_mf_allgather", "slr_reconstruct_mf_allgather_ex", "slr_allgather_clouds", "slr_hybrid_rectify_decode_pair", "slr_reconstruct_hybrid_batch",
    "slr_prefix_index", "slr_compact_points", "slr_cloud_checksums", "slr_verify_assembled",
    "slr_host_alloc", "slr_host_free",
    "slr_timer_begin", "slr_timer_end", "slr_stream_copy", "slr_stream_mix", "slr_profile_enable", "slr_profile_reset",
    "slr_profile_kernel_count", "slr_profile_kernel_name", "slr_profile_get",
]


class SlrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("slr status %d: %s" % (status, message))
        self.status = status


class Camera(C.Structure):
    _fields_ = [("fc", C.c_float * 2), ("cc", C.c_float * 2), ("k", C.c_float * 5),
                ("R", C.c_float * 9), ("t", C.c_float * 3)]


class BatchDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mode", "n_frames", "planes_per_cam", "pitch", "W", "H", "black_thr", "white_thr",
                                        "n_col_bits", "n_row_bits", "scan_w", "scan_h", "rectify", "have_color")]


class RectifyInfo(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("mf_form", C.c_int), ("dma_shape", C.c_int), ("dma_depth", C.c_int),
                ("dma_tiles", C.c_uint), ("dma_nofit_tiles", C.c_uint), ("quads_by_class", C.c_uint * 3),
                ("waves_by_mode", C.c_uint * 3), ("lds_nofit_tiles", C.c_uint * 2), ("dma_extra_entries", C.c_uint)]


MODE_GRAY, MODE_GE, MODE_MF = 0, 1, 2
ASSIGN_CYCLIC, ASSIGN_BLOCKED = 0, 1          # frame -> context assignment of the one-process multi-GPU entries (slr.h)


class Calib(C.Structure):
    _fields_ = [("cam", Camera * 2), ("Q", C.c_double * 16), ("T", C.c_float * 12), ("has_T", C.c_int)]


def make_camera(fc, cc, k, R=None, t=None):
    cam = Camera()
    cam.fc[:] = [float(v) for v in fc]
    cam.cc[:] = [float(v) for v in cc]
    kk = list(k) + [0.0] * (5 - len(k))
    cam.k[:] = [float(v) for v in kk]
    R = np.eye(3) if R is None else np.asarray(R)
    t = np.zeros(3) if t is None else np.asarray(t)
    cam.R[:] = [float(v) for v in R.reshape(-1)]
    cam.t[:] = [float(v) for v in t.reshape(-1)]
    return cam


def make_calib(camL, camR, Q, T=None):
    cal = Calib()
    cal.cam[0] = camL
    cal.cam[1] = camR
    cal.Q[:] = [float(v) for v in np.asarray(Q, np.float64).reshape(-1)]
    if T is not None:
        cal.T[:] = [float(v) for v in np.asarray(T, np.float32).reshape(-1)]
        cal.has_T = 1
    else:
        cal.has_T = 0
    return cal


_lib = None


def load_library():
    """dlopen libslr_hip.so.  torch is imported first so that the HIP runtime already mapped by torch
    (same SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libslr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C structure-light-reconstructor_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    try:
        import torch  # noqa: F401  (plumbing: makes torch's HIP runtime the process-wide one)
    except Exception:  # pragma: no cover - torch is optional for pure C users
        pass
    lib = C.CDLL(LIB_PATH)
    lib.slr_status_string.restype = C.c_char_p
    lib.slr_last_error.restype = C.c_char_p
    lib.slr_profile_kernel_name.restype = C.c_char_p
    lib.slr_last_error.argtypes = [C.c_void_p]
    _lib = lib
    return lib


# Test hook (tests/conftest.py sets it for the whole GPU suite; SLR_POISON_OUTPUTS=1 does the same): every output this module
# allocates is filled with 0x7B bytes before the library sees it, so a pixel no kernel writes cannot pass for a correct one
# because the caching allocator handed out the block of an earlier, correct run (that is how the unwritten parts of split
# tiles stayed hidden in round 3: DESIGN.md section 4).
POISON_OUTPUTS = os.environ.get("SLR_POISON_OUTPUTS", "0") == "1"
# ... and SLR_POISON_SCRATCH=1 makes every Context poison the library's own scratch buffers the same way (SLR_OPT_DEBUG_POISON_SCRATCH)
POISON_SCRATCH = os.environ.get("SLR_POISON_SCRATCH", "0") == "1"


def _empty(shape, dtype, device):
    import torch
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON_OUTPUTS and t.is_cuda:
        t.view(torch.uint8).fill_(0x7B)
        torch.cuda.current_stream(t.device).synchronize()
    return t


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def _mem_of(arrs):
    kinds = set()
    for a in arrs:
        if a is None:
            continue
        if _is_torch(a):
            if not a.is_cuda:
                raise ValueError("torch tensors must live on the GPU (use numpy for host buffers)")
            kinds.add(MEM_DEVICE)
        else:
            kinds.add(MEM_HOST)
    if len(kinds) > 1:
        raise ValueError("all buffers of one call must be host (numpy) or device (torch.cuda)")
    return kinds.pop() if kinds else MEM_HOST


def _ptr(a):
    if a is None:
        return C.c_void_p(None)
    if _is_torch(a):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    assert a.flags.c_contiguous
    return C.c_void_p(a.ctypes.data)


def _plane_ptrs(planes):
    """planes: [N][H][pitch] u8 ndarray/tensor, or a list of N [H][pitch] arrays -> (ptr array, N, H, pitch)."""
    if isinstance(planes, (list, tuple)):
        n = len(planes)
        H, pitch = planes[0].shape
        arr = (C.c_void_p * n)()
        for i, p in enumerate(planes):
            assert tuple(p.shape) == (H, pitch)
            arr[i] = _ptr(p).value
        return arr, n, H, pitch
    n, H, pitch = planes.shape
    base = _ptr(planes).value
    isz = planes.element_size() if _is_torch(planes) else planes.itemsize
    arr = (C.c_void_p * n)()
    for i in range(n):
        arr[i] = base + i * H * pitch * isz
    return arr, n, H, pitch


def _flat(planes):
    return list(planes) if isinstance(planes, (list, tuple)) else [planes]


class Context:
    """One slr_ctx: one GPU, one HIP stream.  Mirrors the C ABI one to one.

    Stream ordering: device-pointer calls are asynchronous on the ctx stream.  When that stream is a torch stream (the
    default: one is created here; or pass `stream=`), every device-pointer call first makes it wait for torch's current
    stream, so tensors torch has just produced are safe to pass in.  Results must not be read by torch before
    `ctx.synchronize()` (or a `torch.cuda.current_stream().wait_stream(ctx.stream)`) -- the rule of any two HIP streams."""

    def __init__(self, device_id=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.slr_create(C.c_int(device_id), C.byref(h))
        if st != OK:
            raise SlrError(st, self.lib.slr_status_string(st).decode())
        self.h = h
        self.device_id = device_id
        self.stream = None                               # torch.cuda.Stream when known (input ordering, see above)
        if stream is None:
            try:
                import torch
                if torch.cuda.is_available():
                    stream = torch.cuda.Stream(device=device_id)
            except Exception:                            # torch is plumbing, not a requirement of the binding
                stream = None
        if stream is not None:
            self.set_stream(stream)
        if POISON_SCRATCH:
            self.set_option(OPT_DEBUG_POISON_SCRATCH, 1)

    def _mem(self, arrs):
        """host/device mode of a call; device mode: order the ctx stream after torch's current stream"""
        mem = _mem_of(arrs)
        if mem == MEM_DEVICE and self.stream is not None:
            import torch
            self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))
            # the kernels run on the ctx stream, the tensors were allocated on torch's current stream: tell the caching
            # allocator, or a temporary passed inline (ctx.reconstruct_mf(a.cuda(), ...)) is freed on return and its
            # block handed to another current-stream allocation while the kernels are still queued
            for a in arrs:
                if a is not None and _is_torch(a):
                    a.record_stream(self.stream)
        return mem

    # -- plumbing
    def _chk(self, st):
        if st != OK:
            msg = self.lib.slr_status_string(st).decode()
            detail = self.lib.slr_last_error(self.h)
            raise SlrError(st, msg + (" (" + detail.decode() + ")" if detail else ""))

    def close(self):
        if getattr(self, "h", None):
            self.lib.slr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream):
        """stream: raw hipStream_t as int, a torch.cuda.Stream, or None for the ctx-owned stream."""
        raw = getattr(stream, "cuda_stream", stream)
        self._chk(self.lib.slr_set_stream(self.h, C.c_void_p(raw)))
        self.stream = stream if hasattr(stream, "wait_stream") else None

    def set_option(self, option, value):
        self._chk(self.lib.slr_set_option(self.h, C.c_int(option), C.c_int(value)))

    def synchronize(self):
        self._chk(self.lib.slr_synchronize(self.h))

    def _new(self, like_mem, shape, dtype, like=None):
        if like_mem == MEM_DEVICE:
            import torch
            tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
            t = _empty(shape, dtype=tdt, device=like.device)
            if self.stream is not None:
                t.record_stream(self.stream)            # written on the ctx stream (see _mem)
            return t
        return np.empty(shape, dtype)

    # -- configuration
    def set_calibration(self, calib):
        self._chk(self.lib.slr_set_calibration(self.h, C.byref(calib)))

    def set_rectify_maps(self, cam, map_xy, map_frac):
        H, W = map_frac.shape
        mem = self._mem([map_xy, map_frac])
        self._chk(self.lib.slr_set_rectify_maps(self.h, C.c_int(cam), _ptr(map_xy), _ptr(map_frac),
                                                C.c_int(W), C.c_int(H), C.c_int(mem)))

    def init_rectify_maps(self, cam, M, D, R, P, W, H):
        """cv::initUndistortRectifyMap on the device (stereorect.cpp:42-43); installs the maps for `cam`."""
        a = [np.ascontiguousarray(x, np.float64).reshape(n) for x, n in ((M, 9), (D, 5), (R, 9), (P, 12))]
        self._chk(self.lib.slr_init_rectify_maps(self.h, C.c_int(cam), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]),
                                                 C.c_int(W), C.c_int(H)))

    def rectify_info(self, cam):
        """slr_get_rectify_info as a dict: which fused-decode form the installed maps of `cam` select, tile / quad statistics."""
        info = RectifyInfo()
        self._chk(self.lib.slr_get_rectify_info(self.h, C.c_int(cam), C.byref(info)))
        return {"mf_form": info.mf_form, "dma_shape": info.dma_shape, "dma_depth": info.dma_depth, "dma_tiles": info.dma_tiles,
                "dma_nofit_tiles": None if info.dma_nofit_tiles == 0xFFFFFFFF else info.dma_nofit_tiles,
                "quads_by_class": list(info.quads_by_class), "waves_by_mode": list(info.waves_by_mode),
                "lds_nofit_tiles": list(info.lds_nofit_tiles), "dma_extra_entries": info.dma_extra_entries}

    def get_rectify_maps(self, cam, W, H):
        xy = np.empty((H, W, 2), np.int16)
        fr = np.empty((H, W), np.uint16)
        self._chk(self.lib.slr_get_rectify_maps(self.h, C.c_int(cam), _ptr(xy), _ptr(fr), C.c_int(W), C.c_int(H),
                                                C.c_int(MEM_HOST)))
        return xy, fr

    # -- K1
    def remap_u8(self, cam, src, out=None):
        H, W = src.shape
        mem = self._mem([src, out])
        out = self._new(mem, (H, W), np.uint8, src) if out is None else out
        self._chk(self.lib.slr_remap_u8(self.h, C.c_int(cam), _ptr(src), C.c_int(W), _ptr(out), C.c_int(W),
                                        C.c_int(W), C.c_int(H), C.c_int(mem)))
        return out

    # -- K2
    def mf_decode(self, planes, black_thr, W=None, rectify_cam=None, phase=None, valid=None):
        ptrs, n, H, pitch = _plane_ptrs(planes)
        assert n == MF_PLANES
        W = pitch if W is None else W
        mem = self._mem(_flat(planes) + [phase, valid])
        like = _flat(planes)[0]
        phase = self._new(mem, (H, W), np.float32, like) if phase is None else phase
        valid = self._new(mem, (H, W), np.uint8, like) if valid is None else valid
        if rectify_cam is None:
            st = self.lib.slr_mf_decode(self.h, ptrs, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr),
                                        _ptr(phase), _ptr(valid), C.c_int(mem))
        else:
            st = self.lib.slr_mf_rectify_decode(self.h, C.c_int(rectify_cam), ptrs, C.c_int(pitch), C.c_int(W),
                                                C.c_int(H), C.c_int(black_thr), _ptr(phase), _ptr(valid), C.c_int(mem))
        self._chk(st)
        return phase, valid

    def mf_rectify_decode_pair(self, planesL, planesR, black_thr, W=None, want_valid=True, phase=None):
        """slr_mf_rectify_decode_pair: both cameras of a frame (ONE launch when the LDS-tiled fused forms apply).
        want_valid=False: no valid arrays, invalid pixels carry a NaN phase (what the whole-path entries run)."""
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert n == MF_PLANES and n2 == MF_PLANES and (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR))
        like = _flat(planesL)[0]
        ph = [self._new(mem, (H, W), np.float32, like) for _ in range(2)] if phase is None else phase
        vd = [self._new(mem, (H, W), np.uint8, like) if want_valid else None for _ in range(2)]
        self._chk(self.lib.slr_mf_rectify_decode_pair(self.h, pl, pr, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr),
                                                      _ptr(ph[0]), _ptr(vd[0]), _ptr(ph[1]), _ptr(vd[1]), C.c_int(mem)))
        return ph, vd

    # -- build extension (no reference counterpart): n_freq x n_step fp16 decode
    def mfn_decode(self, planes, n_freq, n_step, black_thr, W=None, phase=None, valid=None):
        """planes: [2 + n_freq*n_step][H][pitch] float16 (numpy = host, torch.cuda = device)."""
        ptrs, n, H, pitch = _plane_ptrs(planes)
        assert n == 2 + n_freq * n_step
        first = _flat(planes)[0]
        assert (first.element_size() if _is_torch(first) else first.itemsize) == 2
        W = pitch if W is None else W
        mem = self._mem(_flat(planes) + [phase, valid])
        phase = self._new(mem, (H, W), np.float32, first) if phase is None else phase
        valid = self._new(mem, (H, W), np.uint8, first) if valid is None else valid
        self._chk(self.lib.slr_mfn_decode(self.h, ptrs, C.c_int(n_freq), C.c_int(n_step), C.c_int(pitch), C.c_int(W),
                                          C.c_int(H), C.c_float(black_thr), _ptr(phase), _ptr(valid), C.c_int(mem)))
        return phase, valid

    # -- K3 / K3'
    def rectify_source_rows(self, cam, row0, rows):
        """(src_row0, src_rows): the source rows destination rows [row0, row0 + rows) of `cam` sample through the installed maps"""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.lib.slr_rectify_source_rows(self.h, C.c_int(cam), C.c_int(row0), C.c_int(rows), C.byref(a), C.byref(b)))
        return a.value, b.value

    def mfn_rectify_decode(self, cam, planes, n_freq, n_step, black_thr, W=None, H=None, row0=0, rows=None, src_row0=0, phase=None,
                           valid=None):
        """planes [2 + F*N][src_rows][pitch] float16: source rows [src_row0, src_row0 + src_rows) of an H-row image (H defaults to
        src_rows: the whole frame); decodes destination rows [row0, row0 + rows) through camera `cam`'s maps."""
        ptrs, n, src_rows, pitch = _plane_ptrs(planes)
        assert n == 2 + n_freq * n_step
        W = pitch if W is None else W
        H = src_rows if H is None else H
        rows = H - row0 if rows is None else rows
        mem = self._mem(_flat(planes) + [phase, valid])
        like = _flat(planes)[0]
        phase = self._new(mem, (rows, W), np.float32, like) if phase is None else phase
        valid = self._new(mem, (rows, W), np.uint8, like) if valid is None else valid
        self._chk(self.lib.slr_mfn_rectify_decode(self.h, C.c_int(cam), ptrs, C.c_int(n_freq), C.c_int(n_step), C.c_int(pitch),
                                                  C.c_int(W), C.c_int(H), C.c_float(black_thr), C.c_int(row0), C.c_int(rows),
                                                  C.c_int(src_row0), C.c_int(src_rows), _ptr(phase), _ptr(valid), C.c_int(mem)))
        return phase, valid

    def gray_decode(self, planes, n_col_bits, n_row_bits, black_thr, white_thr, scan_w, scan_h, W=None,
                    rectify_cam=None):
        ptrs, n, H, pitch = _plane_ptrs(planes)
        assert n >= 2 + 2 * n_col_bits + 2 * n_row_bits
        W = pitch if W is None else W
        mem = self._mem(_flat(planes))
        like = _flat(planes)[0]
        cx = self._new(mem, (H, W), np.int32, like)
        cy = self._new(mem, (H, W), np.int32, like) if n_row_bits > 0 else None
        valid = self._new(mem, (H, W), np.uint8, like)
        args = [ptrs, C.c_int(n_col_bits), C.c_int(n_row_bits), C.c_int(pitch), C.c_int(W), C.c_int(H),
                C.c_int(black_thr), C.c_int(white_thr), C.c_int(scan_w), C.c_int(scan_h), _ptr(cx), _ptr(cy),
                _ptr(valid), C.c_int(mem)]
        if rectify_cam is None:
            st = self.lib.slr_gray_decode(self.h, *args)
        else:
            st = self.lib.slr_gray_rectify_decode(self.h, C.c_int(rectify_cam), *args)
        self._chk(st)
        return cx, cy, valid

    # -- K4
    def mf_triangulate(self, phaseL, validL, phaseR, validR, want_match=True, row0=0, image_h=None, xyz=None, has=None):
        """row0 / image_h: the arrays are a band of rows [row0, row0 + H) of an image_h-row image (row-band sharding).
        xyz / has: caller-owned outputs (same memory kind as the inputs)."""
        H, W = phaseL.shape
        mem = self._mem([phaseL, validL, phaseR, validR, xyz, has])
        xyz = self._new(mem, (H, W, 3), np.float32, phaseL) if xyz is None else xyz
        has = self._new(mem, (H, W), np.uint8, phaseL) if has is None else has
        mk = self._new(mem, (H, W), np.int32, phaseL) if want_match else None
        if image_h is None and row0 == 0:
            self._chk(self.lib.slr_mf_triangulate(self.h, _ptr(phaseL), _ptr(validL), _ptr(phaseR), _ptr(validR),
                                                  C.c_int(W), C.c_int(H), _ptr(xyz), _ptr(has), _ptr(mk), C.c_int(mem)))
        else:
            self._chk(self.lib.slr_mf_triangulate_rows(self.h, _ptr(phaseL), _ptr(validL), _ptr(phaseR), _ptr(validR),
                                                       C.c_int(W), C.c_int(H if image_h is None else image_h), C.c_int(row0),
                                                       C.c_int(H), _ptr(xyz), _ptr(has), _ptr(mk), C.c_int(mem)))
        return xyz, has, mk

    # -- K5
    def ge_triangulate(self, codeL, validL, codeR, validR, whiteL=None, whiteR=None, want_match=True):
        H, W = codeL.shape
        mem = self._mem([codeL, validL, codeR, validR, whiteL, whiteR])
        xyz = self._new(mem, (H, W, 3), np.float32, codeL)
        has = self._new(mem, (H, W), np.uint8, codeL)
        color = self._new(mem, (H, W), np.uint8, codeL) if whiteL is not None else None
        mk = self._new(mem, (H, W), np.int32, codeL) if want_match else None
        self._chk(self.lib.slr_ge_triangulate(self.h, _ptr(codeL), _ptr(validL), _ptr(codeR), _ptr(validR),
                                              C.c_int(W), C.c_int(H), _ptr(whiteL), _ptr(whiteR), _ptr(xyz),
                                              _ptr(has), _ptr(color), _ptr(mk), C.c_int(mem)))
        return xyz, has, color, mk

    # -- K3' scatter + K6
    def ray_triangulate(self, cxL, cyL, vL, cxR, cyR, vR, scan_w, scan_h):
        H, W = cxL.shape
        mem = self._mem([cxL, cyL, vL, cxR, cyR, vR])
        xyz = self._new(mem, (scan_h, scan_w, 3), np.float32, cxL)
        cnt = self._new(mem, (scan_h, scan_w), np.uint8, cxL)
        self._chk(self.lib.slr_ray_triangulate(self.h, _ptr(cxL), _ptr(cyL), _ptr(vL), _ptr(cxR), _ptr(cyR), _ptr(vR),
                                               C.c_int(W), C.c_int(H), C.c_int(scan_w), C.c_int(scan_h),
                                               _ptr(xyz), _ptr(cnt), C.c_int(mem)))
        return xyz, cnt

    # -- PointCloudImage
    def pointcloud_from_grid(self, xyz, has, scan_w, scan_h, color=None):
        H, W = has.shape
        mem = self._mem([xyz, has, color])
        s = self._new(mem, (scan_h, scan_w, 3), np.float32, xyz)
        c = self._new(mem, (scan_h, scan_w), np.uint8, xyz)
        col = self._new(mem, (scan_h, scan_w), np.uint8, xyz) if color is not None else None
        self._chk(self.lib.slr_pointcloud_from_grid(self.h, _ptr(xyz), _ptr(has), _ptr(color), C.c_int(W), C.c_int(H),
                                                    C.c_int(scan_w), C.c_int(scan_h), _ptr(s), _ptr(c), _ptr(col),
                                                    C.c_int(mem)))
        return s, c, col

    def pointcloud_get(self, pc_sum, pc_count):
        mem = self._mem([pc_sum, pc_count])
        n = int(np.prod(pc_count.shape))
        out = self._new(mem, tuple(pc_count.shape) + (3,), np.float32, pc_sum)
        self._chk(self.lib.slr_pointcloud_get(self.h, _ptr(pc_sum), _ptr(pc_count), C.c_size_t(n), _ptr(out),
                                              C.c_int(mem)))
        return out

    def line_line_intersections(self, p1, p2, v1, v2):
        """slr_line_line_intersections: p1, p2 [3]; v1, v2 [n][3] f32 (host arrays) -> (out [n][3], ok [n])"""
        a = [np.ascontiguousarray(x, np.float32) for x in (p1, p2, v1, v2)]
        n = a[2].shape[0]
        out = np.zeros((n, 3), np.float32); ok = np.zeros(n, np.uint8)
        self._chk(self.lib.slr_line_line_intersections(self.h, C.c_size_t(n), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]),
                                                       _ptr(out), _ptr(ok), C.c_int(MEM_HOST)))
        return out, ok

    # -- whole-path drop-ins
    def reconstruct_mf(self, planesL, planesR, black_thr, rectify, W=None, xyz=None, has=None):
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert n == MF_PLANES and n2 == MF_PLANES and (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR) + [xyz, has])
        like = _flat(planesL)[0]
        xyz = self._new(mem, (H, W, 3), np.float32, like) if xyz is None else xyz
        has = self._new(mem, (H, W), np.uint8, like) if has is None else has
        self._chk(self.lib.slr_reconstruct_mf(self.h, pl, pr, C.c_int(pitch), C.c_int(W), C.c_int(H),
                                              C.c_int(black_thr), C.c_int(1 if rectify else 0), _ptr(xyz), _ptr(has),
                                              C.c_int(mem)))
        return xyz, has

    # -- BASELINE config 3: Gray code + phase from one hybrid stack
    def hybrid_rectify_decode_pair(self, planesL, planesR, n_col_bits, black_thr, white_thr, scan_w, W=None):
        """slr_hybrid_rectify_decode_pair: planes [2 + 2 n_col_bits + 12][H][pitch] per camera -> ([code_xL, code_xR], [phaseL, phaseR]);
        code -1 / phase NaN where invalid."""
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert n >= 2 + 2 * n_col_bits + 12 and n2 == n and (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR))
        like = _flat(planesL)[0]
        cx = [self._new(mem, (H, W), np.int32, like) for _ in range(2)]
        ph = [self._new(mem, (H, W), np.float32, like) for _ in range(2)]
        self._chk(self.lib.slr_hybrid_rectify_decode_pair(self.h, pl, pr, C.c_int(n_col_bits), C.c_int(pitch), C.c_int(W), C.c_int(H),
                                                          C.c_int(black_thr), C.c_int(white_thr), C.c_int(scan_w), _ptr(cx[0]), _ptr(ph[0]),
                                                          _ptr(cx[1]), _ptr(ph[1]), C.c_int(mem)))
        return cx, ph

    def reconstruct_hybrid_batch(self, stack, n_col_bits, black_thr, white_thr, scan_w, W=None, xyz=None, has=None, want_codes=False):
        """slr_reconstruct_hybrid_batch: stack = torch.cuda u8 [n_frames][2][planes_per_cam][H][pitch] -> (xyz, has, code_x or None)."""
        import torch
        nf, two, ppc, H, pitch = stack.shape
        assert two == 2 and stack.is_cuda and stack.is_contiguous()
        W = pitch if W is None else W
        xyz = _empty((nf, H, W, 3), dtype=torch.float32, device=stack.device) if xyz is None else xyz
        has = _empty((nf, H, W), dtype=torch.uint8, device=stack.device) if has is None else has
        codes = _empty((nf, 2, H, W), dtype=torch.int32, device=stack.device) if want_codes else None
        self._mem([stack, xyz, has, codes])
        self._chk(self.lib.slr_reconstruct_hybrid_batch(self.h, C.c_int(nf), _ptr(stack), C.c_int(ppc), C.c_int(n_col_bits), C.c_int(pitch),
                                                        C.c_int(W), C.c_int(H), C.c_int(black_thr), C.c_int(white_thr), C.c_int(scan_w),
                                                        _ptr(xyz), _ptr(has), _ptr(codes)))
        return xyz, has, codes

    def reconstruct_mf_cloud(self, planesL, planesR, black_thr, rectify, scan_w, scan_h, W=None):
        """slr_reconstruct_mf_cloud: the whole MF path + the PointCloudImage adaptor -> (pc_sum [scan_h][scan_w][3], pc_count)."""
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert n == MF_PLANES and n2 == MF_PLANES and (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR))
        like = _flat(planesL)[0]
        s = self._new(mem, (scan_h, scan_w, 3), np.float32, like)
        c = self._new(mem, (scan_h, scan_w), np.uint8, like)
        self._chk(self.lib.slr_reconstruct_mf_cloud(self.h, pl, pr, C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr),
                                                    C.c_int(1 if rectify else 0), C.c_int(scan_w), C.c_int(scan_h), _ptr(s), _ptr(c),
                                                    C.c_int(mem)))
        return s, c

    def reconstruct_mf_batch(self, stack, black_thr, rectify, W=None, xyz=None, has=None):
        """stack: torch.cuda u8 [n_frames][2][14][H][pitch]."""
        import torch
        nf, two, n, H, pitch = stack.shape
        assert two == 2 and n == MF_PLANES and stack.is_cuda and stack.is_contiguous()
        W = pitch if W is None else W
        xyz = _empty((nf, H, W, 3), dtype=torch.float32, device=stack.device) if xyz is None else xyz
        has = _empty((nf, H, W), dtype=torch.uint8, device=stack.device) if has is None else has
        self._mem([stack, xyz, has])
        self._chk(self.lib.slr_reconstruct_mf_batch(self.h, C.c_int(nf), _ptr(stack), C.c_int(pitch), C.c_int(W),
                                                    C.c_int(H), C.c_int(black_thr), C.c_int(1 if rectify else 0),
                                                    _ptr(xyz), _ptr(has)))
        return xyz, has

    def reconstruct_batch(self, mode, stack, black_thr, white_thr=0, n_col_bits=0, n_row_bits=0, scan_w=0, scan_h=0, rectify=True,
                          have_color=False, W=None, xyz=None, has=None, color=None):
        """slr_reconstruct_batch: stack = torch.cuda u8 [n_frames][2][planes_per_cam][H][pitch]; returns (xyz, has/count, color)."""
        import torch
        nf, two, ppc, H, pitch = stack.shape
        assert two == 2 and stack.is_cuda and stack.is_contiguous()
        W = pitch if W is None else W
        oh, ow = (scan_h, scan_w) if mode == MODE_GRAY else (H, W)
        xyz = _empty((nf, oh, ow, 3), dtype=torch.float32, device=stack.device) if xyz is None else xyz
        has = _empty((nf, oh, ow), dtype=torch.uint8, device=stack.device) if has is None else has
        if have_color and color is None:
            color = _empty((nf, H, W), dtype=torch.uint8, device=stack.device)
        self._mem([stack, xyz, has, color])
        d = BatchDesc(mode, nf, ppc, pitch, W, H, black_thr, white_thr, n_col_bits, n_row_bits, scan_w, scan_h,
                      1 if rectify else 0, 1 if have_color else 0)
        self._chk(self.lib.slr_reconstruct_batch(self.h, C.byref(d), _ptr(stack), _ptr(xyz), _ptr(has), _ptr(color)))
        return xyz, has, color

    def reconstruct_ge(self, planesL, planesR, n_col_bits, black_thr, white_thr, scan_w, rectify, have_color, W=None):
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert n >= 2 + 2 * n_col_bits and n2 >= 2 + 2 * n_col_bits and (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR))
        like = _flat(planesL)[0]
        xyz = self._new(mem, (H, W, 3), np.float32, like)
        has = self._new(mem, (H, W), np.uint8, like)
        color = self._new(mem, (H, W), np.uint8, like) if have_color else None
        self._chk(self.lib.slr_reconstruct_ge(self.h, pl, pr, C.c_int(n_col_bits), C.c_int(pitch), C.c_int(W),
                                              C.c_int(H), C.c_int(black_thr), C.c_int(white_thr), C.c_int(scan_w),
                                              C.c_int(1 if rectify else 0), C.c_int(1 if have_color else 0),
                                              _ptr(xyz), _ptr(has), _ptr(color), C.c_int(mem)))
        return xyz, has, color

    def reconstruct_gray(self, planesL, planesR, n_col_bits, n_row_bits, black_thr, white_thr, scan_w, scan_h, W=None):
        pl, n, H, pitch = _plane_ptrs(planesL)
        pr, n2, H2, pitch2 = _plane_ptrs(planesR)
        assert (H, pitch) == (H2, pitch2)
        W = pitch if W is None else W
        mem = self._mem(_flat(planesL) + _flat(planesR))
        like = _flat(planesL)[0]
        xyz = self._new(mem, (scan_h, scan_w, 3), np.float32, like)
        cnt = self._new(mem, (scan_h, scan_w), np.uint8, like)
        self._chk(self.lib.slr_reconstruct_gray(self.h, pl, pr, C.c_int(n_col_bits), C.c_int(n_row_bits),
                                                C.c_int(pitch), C.c_int(W), C.c_int(H), C.c_int(black_thr),
                                                C.c_int(white_thr), C.c_int(scan_w), C.c_int(scan_h), _ptr(xyz),
                                                _ptr(cnt), C.c_int(mem)))
        return xyz, cnt

    # -- ordered prefix index / compaction
    def prefix_index(self, flags, column_major=False, first=0, none=0xFFFFFFFF):
        """slr_prefix_index of a [h][w] u8 flag image -> (index [h][w] uint32, total)."""
        h, w = flags.shape
        mem = self._mem([flags])
        if mem == MEM_DEVICE:
            import torch
            idx = _empty((h, w), dtype=torch.int32, device=flags.device)
            tot = torch.zeros(1, dtype=torch.int32, device=flags.device)
            idx.record_stream(self.stream) if self.stream is not None else None
            tot.record_stream(self.stream) if self.stream is not None else None
        else:
            idx, tot = np.empty((h, w), np.uint32), np.zeros(1, np.uint32)
        self._chk(self.lib.slr_prefix_index(self.h, _ptr(flags), C.c_int(w), C.c_int(h), C.c_int(1 if column_major else 0),
                                            C.c_uint32(first), C.c_uint32(none), _ptr(idx), _ptr(tot), C.c_int(mem)))
        if mem == MEM_DEVICE:
            self.synchronize()
            return idx, int(tot.item()) & 0xFFFFFFFF
        return idx, int(tot[0])

    def compact_points(self, xyz, has):
        """slr_compact_points -> (points [count][3], source positions [count])."""
        n = int(np.prod(has.shape))
        mem = self._mem([xyz, has])
        out = self._new(mem, (n, 3), np.float32, xyz)
        src = self._new(mem, (n,), np.int32, xyz)
        if mem == MEM_DEVICE:
            import torch
            cnt = torch.zeros(1, dtype=torch.int32, device=xyz.device)
        else:
            cnt = np.zeros(1, np.uint32)
        self._chk(self.lib.slr_compact_points(self.h, _ptr(xyz), _ptr(has), C.c_size_t(n), _ptr(out), _ptr(src), _ptr(cnt),
                                              C.c_int(mem)))
        if mem == MEM_DEVICE:
            self.synchronize()
        k = int(cnt[0]) & 0xFFFFFFFF
        return out[:k], src[:k]

    # -- measurement
    def cloud_checksums(self, xyz, has):
        """one 64-bit word per frame of a device-resident cloud xyz [n][H][W][3] / has [n][H][W]"""
        nf, H, W = has.shape
        self._mem([xyz, has])
        out = np.zeros(nf, np.uint64)
        self._chk(self.lib.slr_cloud_checksums(self.h, C.c_int(nf), C.c_int(W), C.c_int(H), _ptr(xyz), _ptr(has), _ptr(out)))
        return out

    def stream_copy(self, dst, src):
        """float4 non-temporal device copy on the ctx stream (the box's streaming rate; asynchronous)"""
        nbytes = src.numel() * src.element_size()
        assert dst.numel() * dst.element_size() == nbytes
        self._mem([dst, src])
        self._chk(self.lib.slr_stream_copy(self.h, _ptr(dst), _ptr(src), C.c_size_t(nbytes)))

    def stream_mix(self, dst, src, reads):
        """the copy kernel with a read : write mix: src holds `reads` streams of dst's size, every 16-byte word of dst = their sum"""
        nbytes = dst.numel() * dst.element_size()
        assert src.numel() * src.element_size() == reads * nbytes
        self._mem([dst, src])
        self._chk(self.lib.slr_stream_mix(self.h, _ptr(dst), _ptr(src), C.c_size_t(nbytes), C.c_int(reads)))

    def timer_begin(self):
        self._chk(self.lib.slr_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float(0)
        self._chk(self.lib.slr_timer_end(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        self._chk(self.lib.slr_profile_enable(self.h, C.c_int(1 if on else 0)))

    def profile_reset(self):
        self._chk(self.lib.slr_profile_reset(self.h))

    def profile(self):
        """{kernel name: (total_ms, launches)} for every kernel launched since the last reset."""
        out = {}
        for i in range(self.lib.slr_profile_kernel_count()):
            ms, n = C.c_double(0), C.c_long(0)
            self._chk(self.lib.slr_profile_get(self.h, C.c_int(i), C.byref(ms), C.byref(n)))
            if n.value:
                out[self.lib.slr_profile_kernel_name(C.c_int(i)).decode()] = (ms.value, n.value)
        return out


def _sync_devices(tensors):
    """the ctx streams are not torch streams: everything torch has queued on ANY stream of EVERY device involved (inputs being
    produced, reused allocator blocks still in use) must be complete before the library reads or writes those buffers"""
    import torch
    for d in sorted({t.device.index for t in tensors if t is not None}):
        torch.cuda.synchronize(d)


def _share(nf, n, k, assignment):
    """frames of context k of n: their count (cyclic: f % n == k; blocked: [k S, (k + 1) S), S = ceil(nf / n))"""
    if assignment == ASSIGN_BLOCKED:
        S = (nf + n - 1) // n
        return max(0, min(nf, (k + 1) * S) - min(nf, k * S))
    return (nf - k + n - 1) // n


def _multi_args(ctxs, stacks, assignment=ASSIGN_CYCLIC):
    n = len(ctxs)
    nf = sum(int(s.shape[0]) for s in stacks)
    _, two, npl, H, pitch = stacks[0].shape
    assert two == 2 and npl == MF_PLANES and len(stacks) == n
    for k, s in enumerate(stacks):
        assert s.is_cuda and s.is_contiguous() and int(s.shape[0]) == _share(nf, n, k, assignment)
        assert s.device.index == ctxs[k].device_id, "stacks[k] must live on ctxs[k]'s device"
    return n, nf, H, pitch


def reconstruct_mf_multi(ctxs, stacks, black_thr, rectify, W=None, gather_ctx=0, xyz=None, has=None):
    """slr_reconstruct_mf_multi: one Context per device (or several on one device), stacks[k] = torch.cuda u8
    [frames of k][2][14][H][pitch] on ctxs[k]'s device; frame f of the job is stacks[f % n][f // n].
    xyz / has (optional): per-context output tensors [frames of k][H][W][3] / [frames of k][H][W] on ctxs[k]'s device -- e.g.
    views of each context's assembled arrays, so that slr_allgather_clouds can exchange them in place afterwards.
    Returns (xyz_all, has_all) on ctxs[gather_ctx]'s device (or the per-ctx lists when gather_ctx < 0)."""
    import torch
    n, nf, H, pitch = _multi_args(ctxs, stacks)
    W = pitch if W is None else W
    if xyz is None:
        xyz = [_empty((int(s.shape[0]), H, W, 3), dtype=torch.float32, device=s.device) for s in stacks]
    if has is None:
        has = [_empty((int(s.shape[0]), H, W), dtype=torch.uint8, device=s.device) for s in stacks]
    for k, s in enumerate(stacks):
        assert tuple(xyz[k].shape) == (int(s.shape[0]), H, W, 3) and xyz[k].dtype == torch.float32 and xyz[k].is_contiguous()
        assert tuple(has[k].shape) == (int(s.shape[0]), H, W) and has[k].dtype == torch.uint8 and has[k].is_contiguous()
        assert xyz[k].device == s.device and has[k].device == s.device
    xa = ha = None
    if gather_ctx >= 0:
        gdev = stacks[gather_ctx].device
        xa = _empty((nf, H, W, 3), dtype=torch.float32, device=gdev)
        ha = _empty((nf, H, W), dtype=torch.uint8, device=gdev)
    _sync_devices(list(stacks) + list(xyz) + list(has) + [xa, ha])
    arr_c = (C.c_void_p * n)(*[c.h.value for c in ctxs])
    arr_s = (C.c_void_p * n)(*[s.data_ptr() for s in stacks])
    arr_x = (C.c_void_p * n)(*[t.data_ptr() for t in xyz])
    arr_h = (C.c_void_p * n)(*[t.data_ptr() for t in has])
    st = ctxs[0].lib.slr_reconstruct_mf_multi(arr_c, C.c_int(n), C.c_int(nf), arr_s, C.c_int(pitch), C.c_int(W), C.c_int(H),
                                              C.c_int(black_thr), C.c_int(1 if rectify else 0), arr_x, arr_h, C.c_int(gather_ctx),
                                              _ptr(xa), _ptr(ha))
    ctxs[0]._chk(st)
    return (xa, ha) if gather_ctx >= 0 else (xyz, has)


def verify_assembled(ctxs, xyz_all, has_all):
    """slr_verify_assembled: every context checksums its assembled cloud on its own device; returns the number of (context, frame)
    pairs that differ from context 0's (0 = the exchange is proven)."""
    n = len(ctxs)
    nf, H, W = has_all[0].shape
    _sync_devices(list(xyz_all) + list(has_all))
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    xs = (C.c_void_p * n)(*[_ptr(x).value for x in xyz_all])
    hh = (C.c_void_p * n)(*[_ptr(h).value for h in has_all])
    mism = C.c_int(-1)
    ctxs[0]._chk(ctxs[0].lib.slr_verify_assembled(hs, C.c_int(n), C.c_int(nf), C.c_int(W), C.c_int(H), xs, hh, C.byref(mism)))
    return mism.value


def reconstruct_mf_allgather(ctxs, stacks, black_thr, rectify, W=None, require_peer=False, assignment=None, out=None):
    """slr_reconstruct_mf_allgather (assignment None: frame f -> ctxs[f % n]) / slr_reconstruct_mf_allgather_ex (ASSIGN_CYCLIC or
    ASSIGN_BLOCKED: stacks[k] = the frames [k S, (k + 1) S) of the job, computed in groups and pushed as one copy per destination
    and group): as reconstruct_mf_multi, but EVERY device ends with the assembled cloud.  out = (xyz_all list, has_all list) to
    reuse caller-owned assembled arrays.
    Returns (xyz_all[k], has_all[k], peer_direct): per-ctx lists of [n_frames][H][W][3] / [n_frames][H][W] tensors."""
    import torch
    n, nf, H, pitch = _multi_args(ctxs, stacks, ASSIGN_CYCLIC if assignment is None else assignment)
    W = pitch if W is None else W
    if out is None:
        xa = [_empty((nf, H, W, 3), dtype=torch.float32, device=s.device) for s in stacks]
        ha = [_empty((nf, H, W), dtype=torch.uint8, device=s.device) for s in stacks]
    else:
        xa, ha = out
        for k, s in enumerate(stacks):
            assert tuple(xa[k].shape) == (nf, H, W, 3) and tuple(ha[k].shape) == (nf, H, W) and xa[k].is_contiguous() and ha[k].is_contiguous()
            assert xa[k].device == s.device and ha[k].device == s.device
    _sync_devices(list(stacks) + list(xa) + list(ha))
    arr_c = (C.c_void_p * n)(*[c.h.value for c in ctxs])
    arr_s = (C.c_void_p * n)(*[s.data_ptr() for s in stacks])
    arr_x = (C.c_void_p * n)(*[t.data_ptr() for t in xa])
    arr_h = (C.c_void_p * n)(*[t.data_ptr() for t in ha])
    direct = C.c_int(-1)
    if assignment is None:
        st = ctxs[0].lib.slr_reconstruct_mf_allgather(arr_c, C.c_int(n), C.c_int(nf), arr_s, C.c_int(pitch), C.c_int(W), C.c_int(H),
                                                      C.c_int(black_thr), C.c_int(1 if rectify else 0), arr_x, arr_h,
                                                      C.c_int(1 if require_peer else 0), C.byref(direct))
    else:
        st = ctxs[0].lib.slr_reconstruct_mf_allgather_ex(arr_c, C.c_int(n), C.c_int(nf), arr_s, C.c_int(pitch), C.c_int(W), C.c_int(H),
                                                         C.c_int(black_thr), C.c_int(1 if rectify else 0), arr_x, arr_h,
                                                         C.c_int(assignment), C.c_int(1 if require_peer else 0), C.byref(direct))
    ctxs[0]._chk(st)
    return xa, ha, direct.value


def allgather_clouds(ctxs, xyz_all, has_all, assignment=ASSIGN_BLOCKED, require_peer=False):
    """slr_allgather_clouds: the exchange alone.  xyz_all[k] / has_all[k] = context k's assembled arrays ([n_frames][H][W][3] /
    [n_frames][H][W] on its device) which already hold ITS frames in their slots; returns peer_direct."""
    n = len(ctxs)
    nf, H, W = has_all[0].shape
    for k in range(n):
        assert tuple(xyz_all[k].shape) == (nf, H, W, 3) and tuple(has_all[k].shape) == (nf, H, W)
        assert xyz_all[k].is_contiguous() and has_all[k].is_contiguous() and xyz_all[k].device.index == ctxs[k].device_id
    _sync_devices(list(xyz_all) + list(has_all))
    arr_c = (C.c_void_p * n)(*[c.h.value for c in ctxs])
    arr_x = (C.c_void_p * n)(*[t.data_ptr() for t in xyz_all])
    arr_h = (C.c_void_p * n)(*[t.data_ptr() for t in has_all])
    direct = C.c_int(-1)
    ctxs[0]._chk(ctxs[0].lib.slr_allgather_clouds(arr_c, C.c_int(n), C.c_int(nf), C.c_int(W), C.c_int(H), arr_x, arr_h,
                                                   C.c_int(assignment), C.c_int(1 if require_peer else 0), C.byref(direct)))
    return direct.value
