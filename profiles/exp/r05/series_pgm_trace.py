"""round 5: where a PGM scan's time goes in MFReconstruct::runReconstructionSeries (SLR_SERIES_TRACE lines on stderr).  PGM only (no PNG
encoding: the scans are written in seconds).  Run on the GPU box from the repo root: python profiles/exp/r05/series_pgm_trace.py [N]"""
import ctypes as C, importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
slr.capi.load_library()
host = C.CDLL(os.path.join(os.getcwd(), "structure-light-reconstructor_amd", "libslr_host.so"))
import test_host_mirror as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H, SW, SH = 4096, 3000, 1280, 1024
calib, _ = synth.make_calibration(W, H)
d = tempfile.mkdtemp(prefix="slr_series_", dir="/tmp")
p = lambda a: a.ctypes.data_as(C.c_void_p)
t0 = time.perf_counter()
empty = [np.zeros((0, H, W), np.uint8)] * 2
for sn in range(N):
    st = synth.render_mf_stack(W, H, seed=70 + sn, device="cuda").cpu().numpy()
    proj = T._write_project(host, d, synth, W, H, calib, empty, sn=sn)
    for cam, side, pre in ((0, "left", "L"), (1, "right", "R")):
        for i in range(14):
            path = os.path.join(proj, "scan/%s/%d/%s%d.pgm" % (side, sn, pre, i)).encode()
            assert host.duke_imwrite(path, p(np.ascontiguousarray(st[cam][i])), W, H, 0) == 1
    if sn:
        Tm = np.array([[1, 0, 0, 5.0 * sn], [0, 1, 0, -2.0], [0, 0, 1, 0.5]], np.float64)
        host.duke_export_mat(os.path.join(proj, "scan/transfer_mat%d.txt" % sn).encode(), Tm.ctypes.data_as(C.c_void_p), 3, 4)
print("wrote %d scans as PGM in %.1f s" % (N, time.perf_counter() - t0), flush=True)
err = C.create_string_buffer(512)
ext = b".pgm"
host.duke_run_series(proj.encode(), 0, 4, SW, SH, W, H, 40, 0, ext, None, None, None, err, 512)        # warm-up: every input slot and both contexts exist
for rep in range(3):
    t0 = time.perf_counter()
    done = host.duke_run_series(proj.encode(), 0, N, SW, SH, W, H, 40, 0, ext, None, None, None, err, 512)
    t_all = time.perf_counter() - t0
    assert done == N, err.value
    t0 = time.perf_counter()
    done = host.duke_run_series(proj.encode(), 0, N // 2, SW, SH, W, H, 40, 0, ext, None, None, None, err, 512)
    t_half = time.perf_counter() - t0
    assert done == N // 2, err.value
    print("pgm: series of %d %.1f ms, series of %d %.1f ms -> %.2f ms per additional scan" % (N, t_all * 1e3, N // 2, t_half * 1e3, (t_all - t_half) / (N - N // 2) * 1e3), flush=True)
