"""Disassembly of one kernel of a built object: python tools/co_dis.py gemm_glds.o '<demangled substring>' [out.s]; prints a summary of
scratch / barrier / waitcnt / branch lines and MFMA positions (line numbers within the kernel body)."""
import os, re, subprocess, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from co_regs import code_object, demangle, LLVM, ROOT

def kernel_body(obj, sub):
    if not os.path.exists(obj):
        obj = os.path.join(ROOT, "build", "obj", obj)
    tmp, co = code_object(obj)
    try:
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = re.findall(r"^[0-9a-f]+ <(\S+)>:", dis, re.M)
    dm = demangle(names)
    hits = [n for n, d in zip(names, dm) if sub in d]
    assert len(hits) == 1, [d for d in dm if sub in d]
    body = dis[dis.index(f"<{hits[0]}>:"):]
    body = body[:body.index("s_endpgm")]
    return [l.split("//")[0].strip() for l in body.split("\n")]

if __name__ == "__main__":
    lines = kernel_body(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(f"{i:5d} {l}" for i, l in enumerate(lines)))
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    print("lines", len(lines), "mfma", len(mf), "first", mf[0], "last", mf[-1])
    for i, l in enumerate(lines):
        if "scratch_" in l or "s_barrier" in l or ("s_cbranch" in l and int(l.split()[-1]) > 60000) or "buffer_load" in l and False:
            print(i, l)
