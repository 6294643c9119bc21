"""Register / spill / LDS figures of the kernels in libslr_hip.so (code-object metadata), filtered by a substring.
usage: python profiles/kernel_regs.py [substr ...]"""
import glob, os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    subs = sys.argv[1:] or [""]
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(os.path.join(ROOT, "structure-light-reconstructor_amd", "libslr_hip.so"), d)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", "libslr_hip.so"], cwd=d, check=True, capture_output=True)
        for f in glob.glob(os.path.join(d, "*gfx950*")):
            t = subprocess.run([LLVM + "/llvm-readelf", "--notes", f], check=True, capture_output=True, text=True).stdout
            for b in t.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", b).group(1)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                if not any(s in dem for s in subs):
                    continue
                g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
                print("%-120s vgpr %3s sgpr %3s spill %2s lds %6s scratch %4s" % (
                    dem.replace("slr::", "")[:120], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                    g("group_segment_fixed_size"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
