"""Register / spill / scratch metadata of every kernel in a built object file (build/obj/<name>.o), from the gfx950 code object's ELF notes.
usage: python tools/co_regs.py gemm_glds.o [substring]"""
import os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
    return tmp, co


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return names


def kernel_table(obj):
    tmp, co = code_object(obj)
    try:
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))
        rows.append(dict(name=re.search(r"\.name:\s+(\S+)", k).group(1), agpr=int(re.match(r":\s+(\d+)", k).group(1)), vgpr=g("vgpr_count"),
                         spill=g("vgpr_spill_count"), sgpr=g("sgpr_count"), sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size")))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = d
    return rows


if __name__ == "__main__":
    obj = sys.argv[1]
    if not os.path.exists(obj):
        obj = os.path.join(ROOT, "build", "obj", obj)
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in kernel_table(obj):
        if sub in r["demangled"]:
            print(f"vgpr {r['vgpr']:4d} agpr {r['agpr']:4d} spill {r['spill']:4d} sgpr {r['sgpr']:4d} sspill {r['sspill']:3d} scratch {r['scratch']:5d}  {r['demangled'][:160]}")
