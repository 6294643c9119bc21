"""No-GPU checks of the drop-in boundary: libslr_hip.so loads and exports every symbol include/slr.h declares;
without a GPU the product path fails loudly (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "slr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(slr):
    declared = _declared_functions()
    assert len(declared) >= 30
    assert declared == sorted(slr.capi.SYMBOLS)


def test_library_exports_every_declared_symbol(slr):
    lib = slr.capi.load_library()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.slr_version() == 100
    assert lib.slr_status_string(0).decode() == "ok"
    assert b"device" in lib.slr_status_string(slr.capi.ERR_NO_DEVICE)
    names = [lib.slr_profile_kernel_name(i).decode() for i in range(lib.slr_profile_kernel_count())]
    assert "slr_mf_decode" in names and "slr_mf_match_triangulate" in names and all(names)


def test_header_constants_match_the_binding(slr):
    """option ids, status codes and memory kinds of include/slr.h == the ctypes mirror's"""
    text = open(os.path.join(ROOT, "include", "slr.h")).read()
    defines = {k: int(v) for k, v in re.findall(r"#define\s+(SLR_[A-Z0-9_]+)\s+(-?\d+)\b", text)}
    enums = {k: int(v) for k, v in re.findall(r"\b(SLR_[A-Z_]+)\s*=\s*(-?\d+)", text)}
    cap = slr.capi
    for name in ("MF_MATCH_ALGO", "MF_DECODE_VEC", "RECT_DECODE_ALGO", "ASYNC_HOST", "PROFILE_STRIDE", "RECT_DMA_SHAPE",
                 "RECT_DMA_DEPTH", "DEBUG_RECT_RESIDENT", "DEBUG_FLAGS", "DEBUG_K4_STOP", "HYBRID_ONE_PASS", "BATCH_STREAMS", "DEBUG_POISON_SCRATCH",
                 "EVAL_MODEL", "MF_BATCH_GROUP", "MF_BATCH_DECODE_GROUP"):
        assert defines["SLR_OPT_" + name] == getattr(cap, "OPT_" + name), name
    assert sorted(v for k, v in defines.items() if k.startswith("SLR_OPT_")) == list(range(1, 17))   # no duplicate ids
    for name in ("OK", "ERR_INVALID_ARG", "ERR_NO_DEVICE", "ERR_HIP", "ERR_NOT_CONFIGURED", "ERR_UNSUPPORTED", "ERR_OOM"):
        assert enums["SLR_" + name] == getattr(cap, name), name
    assert (enums["SLR_MEM_HOST"], enums["SLR_MEM_DEVICE"]) == (cap.MEM_HOST, cap.MEM_DEVICE)
    assert defines["SLR_MF_PLANES"] == 14


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under the product package may import/link/execute it"""
    pkg = os.path.join(ROOT, "structure-light-reconstructor_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "slr_oracle" not in text and "import oracle" not in text and "oracle/" not in text, f


@pytest.mark.skipif(torch.cuda.is_available(), reason="GPU present")
def test_no_gpu_means_loud_failure(slr):
    with pytest.raises(slr.SlrError) as e:
        slr.Context(0)
    assert e.value.status == slr.capi.ERR_NO_DEVICE


def test_header_is_plain_c_and_links(tmp_path):
    """include/slr.h compiles as strict C99 (and as C++), and a C program that takes the address of every declared
    function links against libslr_hip.so -- what a maintainer of the reference application would do first"""
    import subprocess
    names = _declared_functions()
    src = tmp_path / "link_all.c"
    src.write_text('#include "slr.h"\n#include <stdio.h>\nint main(void)\n{\n    const void *f[] = {\n'
                   + "".join("        (const void *)%s,\n" % n for n in names)
                   + '    };\n    unsigned i, n = 0;\n    for (i = 0; i < sizeof f / sizeof f[0]; i++) n += f[i] != 0;\n'
                     '    printf("%u %d\\n", n, slr_version());\n    return 0;\n}\n')
    pkg = os.path.join(ROOT, "structure-light-reconstructor_amd")
    exe = tmp_path / "link_all"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
           "-o", str(exe), "-L", pkg, "-lslr_hip", "-Wl,-rpath," + pkg]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == len(names) and int(out[1]) == 100
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", "-I",
                    os.path.join(ROOT, "include"), str(src)], check=True, capture_output=True)


def check_ticket_register(lib_path, tmp_path):
    """the invariant of test_ticket_register_of_the_fused_decodes_is_untouched on the library file `lib_path`"""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    lib = tmp_path / "lib.so"
    shutil.copy(lib_path, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)   # unbundles next to the copy
    reg = re.search(r'#define\s+SLR_TICKET_SGPR\s+"s(\d+)"',
                    open(os.path.join(ROOT, "structure-light-reconstructor_amd", "csrc", "kernels_rectdma.hip")).read())
    assert reg, "SLR_TICKET_SGPR"
    n = int(reg.group(1))
    # the register by itself, or inside a range s[a:b]
    def names_it(ins):
        if re.search(r"\bs%d\b" % n, ins):
            return True
        return any(int(a) <= n <= int(b) for a, b in re.findall(r"s\[(\d+):(\d+)\]", ins))
    allowed = [re.compile(r"^s_mov_b32 s%d, 1$" % n), re.compile(r"^s_atomic_add s%d, s\[\d+:\d+\], 0x0 glc$" % n),
               re.compile(r"^s_mov_b32 s\d+, s%d$" % n)]
    kernels, issues, takes = 0, 0, 0
    for f in sorted(os.listdir(tmp_path)):
        if not f.endswith("gfx950"):
            continue
        text = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        if "rect_decode_dma_kernel" not in text:
            continue
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
            if m:
                cur = m.group(1)
                kernels += 1 if "rect_decode_dma_kernel" in cur and not cur.endswith(".kd") else 0
                continue
            if not cur or "rect_decode_dma_kernel" not in cur or "\t" not in line:
                continue
            ins = " ".join(line.split("//")[0].split())
            if not names_it(ins):
                continue
            assert any(p.match(ins) for p in allowed), (cur, ins)
            issues += 1 if ins.startswith("s_atomic_add") else 0
            takes += 1 if allowed[2].match(ins) else 0
    assert kernels >= 6 and issues >= kernels and takes >= kernels, (kernels, issues, takes)
    return kernels


def test_ticket_register_of_the_fused_decodes_is_untouched(tmp_path):
    """The LDS-DMA fused decodes draw the next tile's ticket with a scalar atomic whose result arrives a phase later in a FIXED
    SGPR (kernels_rectdma.hip, SLR_TICKET_SGPR) that the kernels keep out of register allocation -- a compiler-visible register
    was copied before the result had arrived (entries of split tiles never decoded).  The invariant is checked on the device code
    of the library as built: inside every *_rect_decode_dma_kernel that register is written by the issue (s_mov 1, s_atomic_add),
    read by the take (s_mov to another SGPR) and named by nothing else.  (tests/test_gpu_disasm.py runs the same check, under
    -m gpu, on the library file the GPU box's test process has actually mapped.)"""
    check_ticket_register(os.path.join(ROOT, "structure-light-reconstructor_amd", "libslr_hip.so"), tmp_path)


def _lds_inflight_violations(listing, kernel_substr="rect_decode_dma_kernel"):
    """Linear scan of an llvm-objdump listing: the destination of a ds_read is 'in flight' until an s_waitcnt lgkmcnt(n) that
    retires it (LDS operations return in order; with a scalar-memory operation outstanding only lgkmcnt(0) retires anything).
    Returns (number of ds_reads seen, [(kernel, instruction, registers)] of instructions that name an in-flight register)."""
    def regs(tok):
        out = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
            out.update(range(int(a), int(b) + 1))
        out.update(int(a) for a in re.findall(r"\bv(\d+)\b", tok))
        return out
    cur, pend, viol, reads = None, [], [], 0
    for line in listing.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur, pend = m.group(1), []
            continue
        if not cur or kernel_substr not in cur or "\t" not in line:
            continue
        ins = " ".join(line.split("//")[0].split())
        if not ins:
            continue
        mn, _, ops = ins.partition(" ")
        if mn == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", ops)
            n = 0 if ops.strip() == "0" else int(mm.group(1)) if mm else None
            if n == 0:
                pend = []
            elif n is not None and not any(k == "smem" for k, _ in pend):
                del pend[:max(0, len(pend) - n)]
            continue
        inflight = set().union(*[d for k, d in pend if k == "ds"]) if pend else set()
        if mn.startswith("ds_read"):
            first, _, rest = ops.partition(",")
            if regs(rest) & inflight:
                viol.append((cur, ins, sorted(regs(rest) & inflight)))
            pend.append(("ds", regs(first)))
            reads += 1
            continue
        if regs(ops) & inflight:
            viol.append((cur, ins, sorted(regs(ops) & inflight)))
        if mn.startswith("ds_"):
            pend.append(("dsw", set()))
        elif mn.startswith(("s_load", "s_atomic", "s_buffer_load", "s_store", "s_dcache")):
            pend.append(("smem", set()))
    return reads, viol


def check_lds_reads_in_flight(lib_path, tmp_path):
    """the invariant of test_lds_reads_in_flight_are_not_touched on the library file `lib_path`"""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    lib = tmp_path / "lib.so"
    shutil.copy(lib_path, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, check=True, capture_output=True)
    reads = 0
    for f in sorted(os.listdir(tmp_path)):
        if not f.endswith("gfx950"):
            continue
        text = subprocess.run([objdump, "-d", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        if "rect_decode_dma_kernel" not in text:
            continue
        n, viol = _lds_inflight_violations(text)
        assert not viol, viol[:5]
        reads += n
    assert reads > 5000, reads          # (the default build: ~19 000 tap reads over 18 kernel instances)
    return reads


def test_lds_reads_in_flight_are_not_touched(tmp_path):
    """The tap reads of the LDS-DMA fused decodes are inline-asm ds_read_b32 issued a pixel ahead of their use and retired by
    COUNTED s_waitcnt lgkmcnt (kernels_rectdma.hip, dma_rd / dma_rd_wait).  Between the two statements the compiler believes the
    destination registers hold their values -- the same hazard class as the in-flight ticket (a copy, a spill or a reordered use
    would read the register before the LDS has written it; gfx950 has no interlock for that).  Checked on the device code of the
    library as built: no instruction of any *_rect_decode_dma_kernel names a register with an LDS read outstanding."""
    import shutil
    import subprocess
    # the checker itself, on a planted violation and on its repaired twin
    bad = "0000 <k_rect_decode_dma_kernel>:\n\tds_read_b32 v1, v9 offset:16\n\tds_read_b32 v2, v9 offset:20\n\ts_waitcnt lgkmcnt(1)\n" \
          "\tv_add_u32_e32 v3, v1, v2\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v4, v2\n"
    n, v = _lds_inflight_violations(bad)
    assert n == 2 and len(v) == 1 and v[0][2] == [2], v
    assert _lds_inflight_violations(bad.replace("lgkmcnt(1)", "lgkmcnt(0)"))[1] == []
    check_lds_reads_in_flight(os.path.join(ROOT, "structure-light-reconstructor_amd", "libslr_hip.so"), tmp_path)


def test_the_suite_runs_with_poisoned_buffers(slr):
    """tests/conftest.py turns both poison hooks on before the binding is imported: device outputs the wrapper allocates and the
    library's scratch buffers start every call as 0x7B bytes (capi.POISON_OUTPUTS / POISON_SCRATCH -> SLR_OPT_DEBUG_POISON_SCRATCH);
    bench.py removes both from its environment."""
    assert slr.capi.POISON_OUTPUTS and slr.capi.POISON_SCRATCH
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ.pop("SLR_POISON_OUTPUTS", None)' in src and 'os.environ.pop("SLR_POISON_SCRATCH", None)' in src
