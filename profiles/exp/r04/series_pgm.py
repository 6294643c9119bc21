"""row f1 at size, round 4: the series loader (MFReconstruct::runReconstructionSeries through slr_cli's path) on the SAME scans
stored as PNG (zlib inflate on the host CPUs) and as binary PGM (no codec: file -> page-locked memory): ms per scan in the steady
state.  Run on the GPU box from the repo root: python profiles/exp/r04/series_pgm.py [N]"""
import ctypes as C, importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
slr.capi.load_library()
host = C.CDLL(os.path.join(os.getcwd(), "structure-light-reconstructor_amd", "libslr_host.so"))
import test_host_mirror as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
W, H, SW, SH = 4096, 3000, 1280, 1024
calib, _ = synth.make_calibration(W, H)
d = tempfile.mkdtemp(prefix="slr_series_", dir="/tmp")
p = lambda a: a.ctypes.data_as(C.c_void_p)
t0 = time.perf_counter()
for sn in range(N):
    st = synth.render_mf_stack(W, H, seed=70 + sn, device="cuda").cpu().numpy()
    proj = T._write_project(host, d, synth, W, H, calib, st, sn=sn)
    for cam, side, pre in ((0, "left", "L"), (1, "right", "R")):
        for i in range(14):
            path = os.path.join(proj, "scan/%s/%d/%s%d.pgm" % (side, sn, pre, i)).encode()
            assert host.duke_imwrite(path, p(np.ascontiguousarray(st[cam][i])), W, H, 0) == 1
    if sn:
        Tm = np.array([[1, 0, 0, 5.0 * sn], [0, 1, 0, -2.0], [0, 0, 1, 0.5]], np.float64)
        host.duke_export_mat(os.path.join(proj, "scan/transfer_mat%d.txt" % sn).encode(), Tm.ctypes.data_as(C.c_void_p), 3, 4)
print("wrote %d scans as PNG and PGM in %.1f s" % (N, time.perf_counter() - t0))
err = C.create_string_buffer(512)
res = {}
for ext in (b".png", b".pgm", b".png", b".pgm"):
    ss = np.zeros((N, SH, SW, 3), np.float32); sc = np.zeros((N, SH, SW), np.uint8)
    host.duke_run_series(proj.encode(), 0, 1, SW, SH, W, H, 40, 0, ext, None, None, None, err, 512)        # warm-up
    t0 = time.perf_counter()
    done = host.duke_run_series(proj.encode(), 0, N, SW, SH, W, H, 40, 0, ext, None, p(ss), p(sc), err, 512)
    t_all = time.perf_counter() - t0
    assert done == N, err.value
    t0 = time.perf_counter()
    host.duke_run_series(proj.encode(), 0, 2, SW, SH, W, H, 40, 0, ext, None, None, None, err, 512)
    t_two = time.perf_counter() - t0
    per = (t_all - t_two) / (N - 2) * 1e3
    print("%s: series of %d %.1f ms, series of 2 %.1f ms -> %.1f ms per additional scan" % (ext.decode(), N, t_all * 1e3, t_two * 1e3, per))
    res.setdefault(ext, []).append((per, sc.copy(), ss.copy()))
assert np.array_equal(res[b".png"][0][1], res[b".pgm"][0][1]) and np.array_equal(res[b".png"][0][2].view(np.uint32), res[b".pgm"][0][2].view(np.uint32))
print("clouds of the PNG and the PGM series identical: True; cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
