"""Per kernel of a built object: counts of global load / store instructions by width and of waits inside; flags kernels whose global loads are
mostly 4-byte (a `cond ? *(float4*)p : zero` select that the compiler split into four predicated dword loads - round 4 finding).
usage: python tools/co_loads.py [obj.o ...]   (default: every object under build/obj)"""
import glob, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from co_regs import code_object, demangle, LLVM, ROOT
import shutil

def kernels(obj):
    tmp, co = code_object(obj)
    try:
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = []
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\s*$|\Z)", dis, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "s_endpgm" not in body:
            continue
        c = lambda pat: len(re.findall(pat, body))
        out.append(dict(name=name, ld1=c(r"global_load_dword\s"), ld2=c(r"global_load_dwordx2\s"), ld4=c(r"global_load_dwordx4\s"), lds=c(r"global_load_lds"),
                        st1=c(r"global_store_dword\s"), st2=c(r"global_store_dwordx2\s"), st4=c(r"global_store_dwordx4\s"), scr=c(r"scratch_")))
    return out

def main():
  objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.o")))
  for o in objs:
      if not os.path.exists(o):
          o = os.path.join(ROOT, "build", "obj", o)
      try:
          ks = kernels(o)
      except subprocess.CalledProcessError:
          continue  # host-only object: no device code
      names = demangle([k["name"] for k in ks])
      for k, n in zip(ks, names):
          flag = "  <-- mostly 4-byte loads" if k["ld1"] >= 8 and k["ld1"] > 2 * (k["ld4"] + k["ld2"]) else ""
          n = n.replace("(anonymous namespace)::", "").replace("void ", "")
          n = re.sub(r"\(.*", "", n)
          print(f"{os.path.basename(o):18s} ld 1/2/4 {k['ld1']:4d} {k['ld2']:4d} {k['ld4']:4d}  dma {k['lds']:4d}  st 1/2/4 {k['st1']:4d} {k['st2']:4d} {k['st4']:4d}  scratch {k['scr']:4d}  {n[:90]}{flag}")

if __name__ == "__main__":
    main()
