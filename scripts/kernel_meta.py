"""Register / LDS / spill figures of every kernel in libtedspad_hip.so (llvm-readelf --notes of the gfx950 code objects): python scripts/kernel_meta.py [substring ...]"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "lib.so")
shutil.copy(os.path.join(ROOT, "ted_spad_amd", "libtedspad_hip.so"), so)
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", so], cwd=tmp, check=True, capture_output=True)
rows = []
for co in glob.glob(os.path.join(tmp, "lib.so.*gfx950*")):
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))
        rows.append((re.search(r"\.name:\s+(\S+)", k).group(1), g("vgpr_count"), int(re.match(r":\s+(\d+)", k).group(1)), g("sgpr_count"),
                     g("group_segment_fixed_size"), g("vgpr_spill_count"), g("private_segment_fixed_size")))
shutil.rmtree(tmp, ignore_errors=True)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%5s %5s %5s %7s %5s %7s  kernel" % ("vgpr", "agpr", "sgpr", "lds", "spill", "scratch"))
for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
    if len(sys.argv) > 1 and not any(s in n for s in sys.argv[1:]):
        continue
    print("%5d %5d %5d %7d %5d %7d  %s" % (r[1], r[2], r[3], r[4], r[5], r[6], n[:150]))
