"""row f1 at size: a 4096x3000 project directory with N multi-frequency scans (28 PNG files each) through slr_cli's series path
(MFReconstruct::runReconstructionSeries: PNG inflate into page-locked memory by a thread pool, two contexts with
SLR_OPT_ASYNC_HOST) vs one scan at a time.  Run on the GPU box from the repo root: python profiles/exp/series_time.py [N]"""
import ctypes as C, importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
slr.capi.load_library()
host = C.CDLL(os.path.join(os.getcwd(), "structure-light-reconstructor_amd", "libslr_host.so"))
import test_host_mirror as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 3           # the series is run over the N scans REP times in a row (same files)
W, H, SW, SH = 4096, 3000, 1280, 1024
calib, _ = synth.make_calibration(W, H)
d = tempfile.mkdtemp(prefix="slr_series_", dir="/tmp")
t0 = time.perf_counter()
for sn in range(N):
    st = synth.render_mf_stack(W, H, seed=70 + sn, device="cuda").cpu().numpy()
    proj = T._write_project(host, d, synth, W, H, calib, st, sn=sn)
    if sn:
        Tm = np.array([[1, 0, 0, 5.0 * sn], [0, 1, 0, -2.0], [0, 0, 1, 0.5]], np.float64)
        host.duke_export_mat(os.path.join(proj, "scan/transfer_mat%d.txt" % sn).encode(), Tm.ctypes.data_as(C.c_void_p), 3, 4)
print("wrote %d scans (%d PNG files) in %.1f s" % (N, 28 * N, time.perf_counter() - t0))
err = C.create_string_buffer(512)
ss = np.zeros((N, SH, SW, 3), np.float32); sc = np.zeros((N, SH, SW), np.uint8)
p = lambda a: a.ctypes.data_as(C.c_void_p)
host.duke_run_series(proj.encode(), 0, 1, SW, SH, W, H, 40, 0, b".png", None, None, None, err, 512)        # warm-up: contexts, maps, tables
t0 = time.perf_counter()
done = host.duke_run_series(proj.encode(), 0, N, SW, SH, W, H, 40, 0, b".png", None, p(ss), p(sc), err, 512)
t_series = time.perf_counter() - t0
assert done == N, err.value
t0 = time.perf_counter()
done = host.duke_run_series(proj.encode(), 0, 1, SW, SH, W, H, 40, 0, b".png", None, None, None, err, 512)
t_one = time.perf_counter() - t0                              # fixed cost (contexts, maps, tables, pinned buffers) + one scan
NH = max(N // 2, 4)                                           # a shorter series that already pins every input slot
t0 = time.perf_counter()
done = host.duke_run_series(proj.encode(), 0, NH, SW, SH, W, H, 40, 0, b".png", None, None, None, err, 512)
t_half = time.perf_counter() - t0
assert done == NH, err.value
t0 = time.perf_counter()
for sn in range(N):
    es = np.zeros((SH, SW, 3), np.float32); ec = np.zeros((SH, SW), np.uint8)
    assert host.duke_run_project(proj.encode(), 2, sn, SW, SH, W, H, 40, 0, 0, b".png", None, p(es), p(ec), None, err, 512) == 1, err.value
    assert np.array_equal(ec, sc[sn]) and np.array_equal(es.view(np.uint32), ss[sn].view(np.uint32))
t_single = time.perf_counter() - t0
# decode alone: the same files into pageable memory, thread pool
t0 = time.perf_counter()
buf = np.zeros((H, W), np.uint8); w_, h_ = C.c_int(0), C.c_int(0)
for i in range(14):
    host.duke_imread(os.path.join(proj, "scan/left/0/L%d.png" % i).encode(), p(buf), buf.size, C.byref(w_), C.byref(h_))
t_dec1 = (time.perf_counter() - t0) / 14
print("series of %d: %.1f ms in all, a series of 1: %.1f ms -> %.1f ms per additional scan (steady state of the pipeline)" %
      (N, t_series * 1e3, t_one * 1e3, (t_series - t_one) / (N - 1) * 1e3))
if N > NH:
    print("series of %d: %.1f ms -> %.1f ms per scan between the two series lengths (every slot pinned in both)" %
          (NH, t_half * 1e3, (t_series - t_half) / (N - NH) * 1e3))
print("one by one: %.1f ms per scan (new objects, contexts and maps per scan, as slr_cli without --series)" % (t_single / N * 1e3))
print("PNG inflate of one 4096x3000 plane on one core: %.1f ms -> 28 files / %d cores" % (t_dec1 * 1e3, os.cpu_count()))
print("clouds identical: True; valid cells per scan:", [int((sc[k] > 0).sum()) for k in range(N)])
