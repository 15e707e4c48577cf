"""Scan the gfx950 ISA of one translation unit for VMEM stores whose data registers are overwritten within a few instructions
(write-after-read on store data).  hipcc's hazard recognizer inserts the documented wait state only when the store's scalar
offset is NOT a register; round 4 found a buffer_store_dwordx4 with an SGPR offset followed two instructions later by a VALU
write of its first data register delivering the new value in some lanes on gfx950 (csrc/inter_mfma.hip, store_g).
  python tools/isa_hazards.py epn_pointcloud_amd/csrc/inter_mfma.hip [kernel-name-regex]     (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, src], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    names = [l[:-1].split(":")[0] for l in lines if re.match(r"^_Z\w+:", l)]
    total = 0
    for nm in names:
        if not re.search(pat, nm):
            continue
        start = lines.index([l for l in lines if l.startswith(nm + ":")][0])
        end = start
        while not lines[end].strip().startswith(".Lfunc_end"):
            end += 1
        body = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
        for i, l in enumerate(body):
            if not re.match(r"(buffer|global|flat)_store", l):
                continue
            toks = l.replace(",", " ").split()
            data = regs(toks[1]) if l.startswith("buffer") else regs(toks[2])
            sgpr_off = l.startswith("buffer") and re.search(r"s\[\d+:\d+\], s\d+", l) is not None
            for k in range(i + 1, min(i + 4, len(body))):
                t2 = body[k].replace(",", " ").split()
                if len(t2) > 1 and (regs(t2[1]) & data) and "_store" not in t2[0] and not t2[0].startswith("s_"):
                    total += 1
                    print(f"{nm[:70]}: +{k - i}  {l[:64]}  <-  {body[k][:60]}" + ("   [SGPR soffset]" if sgpr_off else ""))
    print(f"{total} store(s) with a write of their data registers within 3 instructions")


if __name__ == "__main__":
    main()
