"""Scan the gfx950 ISA of every kernel in tensorflow_ocr_amd/csrc for loads that hipcc serialised.

    python3 scripts/isa_scan.py [file.hip ...]          # default: every .hip under tensorflow_ocr_amd/csrc

For each source it runs `hipcc -S --cuda-device-only` and reports, per kernel,
  * `serial`: the number of places where a vector-memory load is followed by `s_waitcnt vmcnt(0)` and then by another
    load within WINDOW lines with no store / barrier in between — the signature of a "batch" of requests that went out
    one at a time because each sat under a branch (hipcc cannot count the outstanding loads across the two
    paths of a branch and waits for all of them at the merge);
  * `loads`, `vm0` (vmcnt(0) waits), `scratch` (scratch loads: spills or a descriptor pinned in private memory),
    `vgpr` / `spill` from the kernel's metadata.
Kernels whose waits are counted by hand (inline-asm main loops) show `serial = 0`."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 40
LOAD = re.compile(r"^\s+(global_load|flat_load|buffer_load|scratch_load)")
STORE = re.compile(r"^\s+(global_store|flat_store|buffer_store|s_barrier)")


def scan(path, out):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "-S", "--cuda-device-only", "-o", asm, path], capture_output=True, text=True)
        if r.returncode != 0:
            print("%s: hipcc failed\n%s" % (path, r.stderr[-400:]), file=sys.stderr)
            return
        text = open(asm).read()
    meta = {}
    for b in text.split("  - .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", b)
        if nm:
            meta[nm.group(1)] = (int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)),
                                 int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)))
    names = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+|\w+):\s*(?:;.*)?$", text, re.M) if m.group(1) in meta]
    for i, (pos, name) in enumerate(names):
        body = text[pos:names[i + 1][0] if i + 1 < len(names) else len(text)].split("\n")
        last, waited, serial, loads, vm0, scratch = None, False, 0, 0, 0, 0
        for ln, l in enumerate(body):
            if LOAD.match(l):
                loads += 1
                scratch += "scratch_load" in l
                if last is not None and waited and ln - last <= WINDOW:
                    serial += 1
                last, waited = ln, False
            elif "s_waitcnt" in l and "vmcnt(0)" in l:
                vm0 += 1
                waited = last is not None
            elif STORE.match(l):
                last, waited = None, False
        try:
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except OSError:
            short = ""
        short = short or re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        out.append((serial, os.path.basename(path), short[-70:], loads, vm0, scratch) + meta[name])


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tensorflow_ocr_amd", "csrc", "*.hip")))
    rows = []
    for f in files:
        scan(f, rows)
    rows.sort(reverse=True)
    print("%6s  %-18s %-70s %5s %4s %7s %4s %5s" % ("serial", "file", "kernel", "loads", "vm0", "scratch", "vgpr", "spill"))
    for r in rows:
        if r[0] or r[5] or r[7]:
            print("%6d  %-18s %-70s %5d %4d %7d %4d %5d" % r)


if __name__ == "__main__":
    main()
