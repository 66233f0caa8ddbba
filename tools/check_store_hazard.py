#!/usr/bin/env python3
"""Static check of the built library for the gfx950 write-data hazard of profiles/r03_anomalies.md (B):

a MUBUF / MTBUF store of more than 64 bits whose `soffset` is an SGPR, followed DIRECTLY by an instruction that writes one of
its data VGPRs.  The ISA asks for one wait state between a > 64-bit VMEM store and a VALU write of its data; LLVM's hazard
recogniser (GCNHazardRecognizer::createsVALUHazard) applies it to buffer stores only when `soffset` is not a register, and on
gfx950 the form with a register needs it too (measured: ~20 % of such stores carried the new value in their first dword).

usage: check_store_hazard.py [libpnyolo.so]   -> exit 1 and a listing if the pattern occurs.  Needs llvm-objdump (ROCm)."""
import os, re, shutil, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|tbuffer_store_format_xyzw?|buffer_store_format_xyzw?)\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+?)[\s,]")
DEST = re.compile(r"^\s*(\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b")


def code_objects(lib, tmp):
    dst = os.path.join(tmp, os.path.basename(lib))
    shutil.copy(lib, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def scan(path):
    txt = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True, check=True).stdout
    found, func, prev = [], "?", None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func, prev = m.group(1), None
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        if prev is not None:
            d = DEST.match(ins)
            if d and not ins.startswith(("buffer_store", "global_store", "flat_store", "ds_write", "scratch_store", "s_", "v_cmp", "v_cmpx")):
                lo, hi = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)),) * 2
                if not (hi < prev[1] or lo > prev[2]):
                    found.append((func, prev[0], ins))
            prev = None
        s = STORE.match(ins)
        if s and re.match(r"^(s\d+|ttmp\d+|m0)$", s.group(5)):
            prev = (ins, int(s.group(2)), int(s.group(3)))
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "pixel-nerf-yolo_amd", "libpnyolo.so")
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        objs = code_objects(os.path.abspath(lib), tmp)
        if not objs:
            print("no gfx950 code object found in", lib)
            return 2
        for o in objs:
            bad += scan(o)
    for func, st, nxt in bad:
        print("%s:\n    %s\n    %s" % (func, st, nxt))
    print("%d wide store(s) with an SGPR soffset directly followed by a write of their data registers" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
