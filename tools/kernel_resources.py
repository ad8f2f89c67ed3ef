"""registers, LDS and spills of every kernel in liblinne_amd.so (from the code object's metadata notes): what bounds the waves per SIMD
usage: python3 tools/kernel_resources.py [path/to/liblinne_amd.so] [filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "linne_amd", "liblinne_amd.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
B = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, d + "/fat.bin"])
    subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + d + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + d + "/dev.co"])
    txt = subprocess.run([B + "llvm-readelf", "--notes", d + "/dev.co"], capture_output=True, text=True).stdout
names, rows = [], []
for e in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    e = ".agpr_count:" + e
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", e) or [None, "?"])[1]
    names.append(g("name")); rows.append([g(k) for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "vgpr_spill_count", "max_flat_workgroup_size")])
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, r in sorted(zip(dem, rows)):
    n = re.sub(r"\((Plan|DecPlan|Rice|Train|TrainArgs).*", "", n)
    if flt and flt not in n:
        continue
    v, a, lds, wg = int(r[0]), int(r[1]), int(r[3]), int(r[5])
    tot = ((v + a + 7) // 8) * 8
    by_reg = min(8, 512 // max(tot, 1))
    wpb = (wg + 63) // 64
    by_lds = (160 * 1024 // lds) * wpb / 4 if lds else 99
    print(f"{n[:60]:60s} vgpr {v:4d} agpr {a:3d} sgpr {r[2]:>3s} lds {lds:6d} spill {r[4]:>4s} wg {wg:5d} | waves/SIMD by regs {by_reg}, by LDS {by_lds:.1f}")
